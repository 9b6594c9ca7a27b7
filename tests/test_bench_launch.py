"""bench.py --gpus N without a launcher starts the N-rank run itself: the command it builds, and that a failing
child run gives a non-zero exit code with the reason on stderr.  CPU only: the children stop at the device-count check."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_command_is_one_process_per_gpu_on_loopback():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(2, ["--gpus", "2", "--steps", "3"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3"]            # the script's own arguments follow it unchanged
    assert 1024 < bench.free_port() < 65536


def test_self_launch_propagates_child_failure():
    """Two ranks on a node without (enough) GPUs: every child exits at the device-count check, the parent returns non-zero and
    prints no JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("node has two GPUs: the child run would succeed")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "has no GPU" in r.stderr and "the run needs 2" in r.stderr
    assert "child run failed" in r.stderr
    assert '"metric"' not in r.stdout


def test_mismatched_world_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
