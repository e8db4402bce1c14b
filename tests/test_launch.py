"""raytracer_amd.launch: `bench.py --gpus N` started without torch.distributed.run becomes the parent of N rank
processes.  CPU test of that launcher with a trivial gloo job: the ranks find each other through the environment it
sets, rank 0's stdout reaches the parent's, and a failing rank's exit status is relayed (also when it dies before the
rendezvous, where the surviving ranks would otherwise wait for it)."""
import json
import os
import subprocess
import sys
import time

from raytracer_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "helpers", "launch_child.py")


def test_two_ranks_meet_and_all_reduce(tmp_path):
    rc = launch.spawn_ranks([sys.executable, CHILD, str(tmp_path), "-1", "0"], 2)
    assert rc == 0
    seen = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert [s["rank"] for s in seen] == [0, 1] and [s["local_rank"] for s in seen] == [0, 1]
    assert all(s["world"] == 2 and s["sum"] == 3.0 and s["master"].startswith("127.0.0.1:") for s in seen)
    assert seen[0]["master"] == seen[1]["master"]


def test_exit_status_of_a_failing_rank_is_relayed(tmp_path):
    assert launch.spawn_ranks([sys.executable, CHILD, str(tmp_path), "1", "3"], 2) == 3


def test_a_rank_that_dies_before_the_rendezvous_does_not_hang_the_job(tmp_path):
    t0 = time.monotonic()
    rc = launch.spawn_ranks([sys.executable, CHILD, str(tmp_path), "1", "5", "early"], 2, grace_s=2.0)
    assert rc == 5 and time.monotonic() - t0 < 60.0


def test_parent_relays_rank0_stdout(tmp_path):
    """Through a real parent process, as bench.py uses it: the JSON line of rank 0 is the parent's stdout."""
    code = ("import sys; sys.path.insert(0, %r); from raytracer_amd import launch; "
            "sys.exit(launch.spawn_ranks([sys.executable, %r, %r, '-1', '0'], 2))" % (ROOT, CHILD, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {"ok": True, "world": 2}


def test_rank_environment():
    env = launch.rank_env(3, 8, 1234, base={})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "1234" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert launch.launched_as_rank(env) and not launch.launched_as_rank({})


def test_bench_without_gpus_exits_with_a_message_not_a_traceback():
    """This container has no GPU: `python bench.py --gpus 2` (self-launched ranks) and `--gpus 1` must say so and
    return non-zero (no CPU fallback), without a Python traceback."""
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("needs a box without GPUs")
    env = {k: v for k, v in os.environ.items() if k not in launch.RANK_ENV}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 2 and "visible GPU" in r.stderr and "Traceback" not in r.stderr and r.stdout.strip() == ""
