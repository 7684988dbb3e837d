"""bench.py's own launcher, without a GPU: `python bench.py --gpus N` must itself become an N-rank
torch.distributed.run job (the round's driver starts it exactly like that), and a launcher whose
world size disagrees with --gpus must be refused before anything touches a device."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert p.returncode != 0
    assert "WORLD_SIZE=3" in p.stderr and "--gpus 2" in p.stderr


def test_gpus_n_starts_n_ranks(monkeypatch):
    bench = _bench_module()
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return Done()

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--backend", "gloo"])
    args = bench.parse()
    with pytest.raises(SystemExit) as e:
        bench.start_ranks_if_asked(args)
    assert e.value.code == 7                      # the job's exit status becomes this program's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    at = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[at + 1:] == ["--gpus", "4", "--steps", "2", "--backend", "gloo"]     # the same arguments, behind the script


def test_one_gpu_or_a_matching_launcher_start_nothing(monkeypatch):
    bench = _bench_module()
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: pytest.fail("no child expected"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench.start_ranks_if_asked(bench.parse())
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    bench.start_ranks_if_asked(bench.parse())
