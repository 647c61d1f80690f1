"""bench.py --gpus N without a launcher around it must start its own N ranks as fresh child processes BEFORE anything in the
parent touches the GPU (the driver's N-GPU command has the same shape as its 1-GPU one)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_importing_bench_does_not_import_torch():
    code = "import sys, importlib.util; s = importlib.util.spec_from_file_location('b', %r); m = importlib.util.module_from_spec(s); s.loader.exec_module(m); print('torch' in sys.modules)" % os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "False"


def test_spawn_ranks_builds_a_torchrun_command_on_loopback(monkeypatch):
    bench = _load_bench()
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert rc == 7                                        # the children's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_spawns_before_touching_torch(monkeypatch):
    bench = _load_bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    called = {}

    def fake_spawn(n, argv):
        called["n"], called["argv"], called["torch_loaded"] = n, argv, "torch" in sys.modules and hasattr(sys.modules["torch"], "_bench_marker")
        return 0

    monkeypatch.setattr(bench, "spawn_ranks", fake_spawn)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert called["n"] == 2 and called["argv"] == ["--gpus", "2", "--steps", "1", "--warmup", "0"]
