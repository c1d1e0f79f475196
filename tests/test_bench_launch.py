"""bench.py's N > 1 entry from a bare shell (no GPU needed): it must start its ranks as a child process
before anything touches the GPU, relay the child's status, and keep fd 1 for the one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bare_env():
    return {k: v for k, v in os.environ.items()
            if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_print_launch_is_the_drivers_command_shape():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "7", "--warmup", "2", "--print-launch"],
                         capture_output=True, text=True, timeout=120, env=_bare_env())
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip())
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], BENCH)
    tail = cmd[cmd.index("--master-port") + 3:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def test_bare_multi_gpu_command_launches_children_and_relays_their_status():
    """Here (no GPU) the ranks die at torch.cuda.set_device: the parent must come back with a non-zero
    status, the children's message on stderr, nothing on stdout — and must not be the old SystemExit."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                          "--no-full-solve", "--no-fft", "--rehearse-shared-gpu"],
                         capture_output=True, text=True, timeout=600, env=_bare_env())
    import torch
    if torch.cuda.is_available():          # on a GPU box the rehearsal itself runs (tests/test_configs_gpu.py)
        assert out.returncode == 0, out.stderr[-2000:]
        assert json.loads(out.stdout)["n_gpus"] == 2
        return
    assert out.returncode != 0
    assert out.stdout.strip() == ""
    assert "launch with torch.distributed.run" not in out.stderr
    assert "ChildFailedError" in out.stderr or "cuda" in out.stderr.lower()
