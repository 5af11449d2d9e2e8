"""`python bench.py --gpus N` without a rank environment must start its own ranks (the driver calls it that way)."""
import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")


def test_launcher_command_is_the_contract_form():
    cmd = bench.launcher_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, 29777, python="python")
    assert cmd[:3] == ["python", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_self_launch_relays_one_json_line_and_the_exit_code(monkeypatch, capfd, tmp_path):
    child = tmp_path / "child.py"
    child.write_text("import sys, json\nprint('RCCL banner')\nprint(json.dumps({'value': 7, 'n_gpus': 2}))\nsys.exit(int(sys.argv[1]))\n")
    monkeypatch.setattr(bench, "launcher_command", lambda argv, n, port, python=None: [sys.executable, str(child), "0"])
    assert bench.self_launch(["--gpus", "2"], 2) == 0
    out = capfd.readouterr()
    lines = [ln for ln in out.out.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"value": 7, "n_gpus": 2}
    assert "RCCL banner" in out.err
    monkeypatch.setattr(bench, "launcher_command", lambda argv, n, port, python=None: [sys.executable, str(child), "3"])
    assert bench.self_launch(["--gpus", "2"], 2) == 3
    assert capfd.readouterr().out.strip() == ""


def test_gpus_gt_visible_devices_fails_before_any_gpu_call(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "64"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "GPU(s) visible" in str(e.value)
