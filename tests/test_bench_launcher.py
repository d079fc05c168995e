"""`python bench.py --gpus N` with no WORLD_SIZE in the environment must start its N ranks itself -- as a CHILD process,
before anything has touched a GPU -- relay rank 0's JSON line and leave with the child's exit status (VERDICT round 4,
item 1: without it SURVEY section 8e cannot be measured by a driver that runs `bench.py --gpus 8` the way it runs
`--gpus 1`).  The launcher branch is driven here with a stub in torch.distributed.run's place (SFMHIP_BENCH_LAUNCHER):
no GPU, no torch import in the parent."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _stub(tmp_path, body):
    p = tmp_path / "stub_launcher.py"
    p.write_text(textwrap.dedent(body))
    return f"{sys.executable} {p}"


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(kw)
    return env


def test_gpus_n_without_world_size_starts_a_child_and_relays_its_line(tmp_path):
    stub = _stub(tmp_path, """
        import json, os, sys
        a = sys.argv[1:]
        # what torch.distributed.run would be given: one node, N ranks, static rendezvous on 127.0.0.1, then bench.py + its flags
        assert a[0] == "--nnodes=1" and a[1] == "--nproc-per-node=2", a
        assert a[2:4] == ["--master-addr", "127.0.0.1"] and a[4] == "--master-port" and 1024 < int(a[5]) < 65536, a
        assert os.path.basename(a[6]) == "bench.py" and a[7:] == ["--gpus", "2", "--lean", "--steps", "3"], a
        assert "WORLD_SIZE" not in os.environ and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
        print(json.dumps({"n_gpus": 2, "rccl_ranks": 2, "argv": a[7:]}))
    """)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--lean", "--steps", "3"], env=_env(SFMHIP_BENCH_LAUNCHER=stub),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # ONE JSON line, the child's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert "starting" in r.stderr and "--nproc-per-node=2" in r.stderr


def test_the_child_s_exit_status_is_the_parent_s(tmp_path):
    stub = _stub(tmp_path, "import sys\nsys.exit(7)\n")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(SFMHIP_BENCH_LAUNCHER=stub), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 7 and not r.stdout.strip()


def test_launcher_command_is_torch_distributed_run_by_default(monkeypatch):
    monkeypatch.delenv("SFMHIP_BENCH_LAUNCHER", raising=False)
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(8, ["--gpus", "8"], port=29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:9] == ["--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29517"]
    assert cmd[9] == BENCH and cmd[10:] == ["--gpus", "8"]


def test_with_world_size_set_the_process_is_a_rank_not_a_launcher(tmp_path):
    """Under a launcher (WORLD_SIZE set) bench.py must not start another one: the stub would leave a marker file."""
    marker = tmp_path / "started"
    stub = _stub(tmp_path, f"open({str(marker)!r}, 'w').close()\n")
    # the rank path needs a GPU, which is absent here: it fails later, at the device -- but never through the launcher
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--lean"],
                       env=_env(SFMHIP_BENCH_LAUNCHER=stub, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert not marker.exists()
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
