"""bench.py's launch contract, the parts that need no GPU: `python bench.py --gpus N` without a launcher
re-executes itself under torch.distributed.run with N ranks; a launcher whose WORLD_SIZE disagrees with --gpus is an error."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_without_launcher_reexecutes_under_torch_distributed_run(monkeypatch):
    bench = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--nnodes=1" in cmd
    # the launcher picks its own free port (--standalone: no bind-close-reuse race here) and rendezvouses on 127.0.0.1
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]  # the ranks see the same command line
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # (only a default: a caller's own setting is kept)


def test_hsa_ipc_setting_of_the_caller_is_kept(monkeypatch):
    bench = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["env"] = env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"


def test_gpus_left_at_its_default_adopts_the_launchers_world_size():
    """`torchrun --nproc-per-node 3 bench.py` with no --gpus: the launcher's WORLD_SIZE is the number of ranks (round-4 advisor); the run then stops at
    the next check -- no GPU here -- instead of at a disagreement."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], env=env, capture_output=True, text=True, timeout=300)
    assert "--gpus" not in out.stderr and "rank(s)" not in out.stderr
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and "needs a GPU" in out.stderr


def test_world_size_exported_without_a_launcher_does_not_block_the_self_launch(monkeypatch):
    """A scheduler that merely exports WORLD_SIZE (no RANK) is not a launcher: `python bench.py --gpus 2` still starts its own ranks."""
    bench = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("WORLD_SIZE", "16")
    monkeypatch.delenv("RANK", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "2"


def test_world_size_disagreeing_with_gpus_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 2" in out.stderr and "3 rank" in out.stderr


def test_one_gpu_never_relaunches(monkeypatch):
    bench = _bench()

    def fake_call(cmd, env=None):
        raise AssertionError("N = 1 must not start a launcher")
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import torch
    if torch.cuda.is_available():
        return  # (on a GPU box the run itself is covered by the -m gpu tests)
    try:
        bench.main()
        raise AssertionError("expected the no-GPU exit")
    except SystemExit as e:
        assert "needs a GPU" in str(e.code)
