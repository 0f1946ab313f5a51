"""bench.py's launcher contract, CPU side: `python bench.py --gpus N` with no launcher environment starts its own
ranks as a CHILD `torch.distributed.run` (the reference is launched one process per GPU from torchrun's environment,
utils/utils.py:335-361) before anything touches a GPU, and hands the child's exit code back."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return e


def test_dry_launch_prints_the_child_command():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "7", "--warmup", "2", "--dry_launch"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]           # the ranks get the same arguments, minus --dry_launch


def test_under_a_launcher_the_process_is_a_rank_and_world_must_match():
    """With RANK / WORLD_SIZE present nothing is spawned; a --gpus that disagrees with WORLD_SIZE is refused before any GPU call."""
    e = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29431", SA_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_self_launch_returns_the_childs_exit_code():
    """No GPU in the build container: every rank of the self-launched child fails loudly (no CPU fallback), and the parent's exit code
    is the launcher's -- non-zero, with no JSON line on stdout."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "1", "--no_cpu_baseline"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        return                                                                       # (on a GPU box this would be a real 2-rank run)
    assert r.returncode != 0
    assert "starting" in r.stderr and "torch.distributed.run" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())
