"""The code path `bench.py --gpus N` runs for N > 1 -- process group, C-ABI shard group, group search, repeat rounds, gloo waits, the
single-process leg behind a barrier, teardown -- rehearsed with ONE rank on the one GPU of the box (`--rehearse-distributed`), and
once more with the exchange over torch.distributed that bench.py falls back to when the library's own RCCL group cannot be
created on every rank (`--rehearse-fallback`).  No 8-GPU node has run this code yet; a hang or an asymmetric collective in it
would cost the scaling measurement."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-distributed", "--rows", "300000", "--steps", "6", "--warmup", "2",
                        "--cpu-seconds", "0"] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    return line


def test_distributed_path_with_one_rank_through_the_c_abi_group():
    l = _run([])
    assert l["n_gpus"] == 1 and l["rccl_ranks"] == 1 and l["exchange"] is None and l["value"] > 0
    assert "RCCL all-gather + merge behind the C ABI" in l["config"]["sharding"]
    assert len(l["synchronous_rounds_before_headline_ms_per_step"]) == 3
    assert l["parity"]["ids_and_distances_bit_identical"] is True
    assert l["single_process_sharded"]["peer"]["results_identical_to_headline"] is True


def test_distributed_path_with_the_torch_distributed_fallback_exchange():
    l = _run(["--rehearse-fallback", "--no-single-process"])
    assert l["rccl_ranks"] == 1 and "fallback" in l["exchange"] and "fallback" in l["config"]["sharding"] and l["value"] > 0
    assert l["parity"]["ids_and_distances_bit_identical"] is True
