"""bench.py's `--gpus N` contract on a box WITHOUT a GPU: whatever the environment says, `--gpus N` with N > 1
either runs N ranks or exits non-zero -- it never prints a line that says n_gpus: 1."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLEAN = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                          "FREUD_BENCH_SHARE_GPU")}


def _bench(args, env):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True,
                          timeout=600)


def test_gpus_2_without_rank_environment_refuses_on_a_node_with_fewer_gpus():
    pr = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], CLEAN)
    assert pr.returncode != 0
    assert '"n_gpus"' not in pr.stdout, pr.stdout
    assert "--gpus 2" in pr.stderr and "GPU" in pr.stderr, pr.stderr[-500:]


@pytest.mark.parametrize("world", ["1", "4"])
def test_gpus_must_match_the_launchers_world_size(world):
    pr = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1"], dict(CLEAN, WORLD_SIZE=world, RANK="0", LOCAL_RANK="0"))
    assert pr.returncode != 0
    assert '"n_gpus"' not in pr.stdout, pr.stdout
    assert "--gpus 8" in pr.stderr and f"WORLD_SIZE={world}" in pr.stderr, pr.stderr[-500:]


def test_self_launch_passes_a_rank_failure_on():
    """The parent spawns the ranks (forced here with the shared-GPU switch, since this box shows no GPU); they fail -- there is no
    GPU and no CPU fallback -- and the parent exits non-zero without a JSON line instead of hanging on the surviving ranks."""
    pr = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], dict(CLEAN, FREUD_BENCH_SHARE_GPU="1"))
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    assert pr.returncode != 0
    assert '"n_gpus"' not in pr.stdout, pr.stdout
    assert "self-launched run failed" in pr.stderr, pr.stderr[-800:]
