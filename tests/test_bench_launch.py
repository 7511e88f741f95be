"""`python bench.py --gpus N` must START N ranks when it is not already one of them (VERDICT r2 missing 2: with WORLD_SIZE unset
the flag used to be ignored and the driver's scaling run would have measured one GPU N times)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _dry(n, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "dry_launch" in l]
    return r, lines


def test_bench_gpus_2_spawns_two_ranks():
    r, lines = _dry(2)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted((l["rank"], l["world"], l["local_rank"]) for l in lines) == [(0, 2, 0), (1, 2, 1)], (lines, r.stderr[-2000:])
    assert all(l["gpus"] == 2 and l["master"].startswith("127.0.0.1:") for l in lines)


def test_bench_gpus_1_and_torchrun_children_do_not_spawn():
    r, lines = _dry(1)
    assert r.returncode == 0 and [(l["rank"], l["world"]) for l in lines] == [(0, 1)]
    # already a rank of somebody's torchrun (the driver's N > 1 form): it is that rank, it launches nothing
    r, lines = _dry(4, {"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode == 0 and [(l["rank"], l["world"], l["local_rank"]) for l in lines] == [(2, 4, 2)]


def test_bench_never_touches_the_gpu_before_it_spawns():
    """The parent of the ranks must not initialise HIP (a process that has may not exec, and would hold a context on GPU 0):
    everything above the spawn in main() is argument parsing."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    head = main[:main.index("sys.exit(spawn_ranks(")]
    for needle in ("torch.cuda", "load_package()", "vdb.build", ".to(device", "set_device"):
        assert needle not in head, needle
