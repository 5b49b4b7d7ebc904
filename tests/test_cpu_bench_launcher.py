"""bench.py --gpus N: the flag must start N ranks (or refuse), never silently measure one GPU (driver contract)."""
import importlib.util
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_flag_starts_that_many_ranks_and_relays_rank0_json(monkeypatch, capsys):
    bench = _bench()
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout='noise\n{"metric": "tiles/sec", "n_gpus": 4}\n')

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert capsys.readouterr().out.strip() == '{"metric": "tiles/sec", "n_gpus": 4}'


def test_failed_rank_makes_the_launcher_fail(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=3, stdout="boom\n"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 3


def test_world_size_must_match_gpus(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert "WORLD_SIZE=4" in str(ei.value.code)
    monkeypatch.setattr(sys, "argv", ["bench.py"])            # one rank asked for, launched under a 4-rank launcher: refuse as well
    with pytest.raises(SystemExit):
        bench.main()


def test_real_two_rank_launch_reaches_both_ranks():
    """No GPU here: every rank must stop at the 'needs an MI355X' check -- which proves two processes were started -- and the
    launcher must report the failure instead of printing a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the gloo rehearsal of the GPU suite")
    assert r.returncode != 0
    assert r.stderr.count("needs an MI355X") == 2 and "2-rank run failed" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
