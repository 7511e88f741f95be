"""A short run of tools/fuzz_parity.py (random shapes, metrics, distributions, tombstones, id orders, masks, k, batch
sizes through both tier configurations, compared bit for bit with each other and with the oracle).  The long runs
(260 cases, all identical) are recorded in DESIGN.md section 8."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_configurations_are_bit_identical_to_the_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--cases", "16", "--seed", "5",
                          "--max-rows", "150000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "ALL 16 CASES OK" in out.stdout


def test_random_hnsw_configurations_equal_the_cpu_restatement():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_hnsw.py"), "--cases", "10", "--seed", "4"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "ALL 10 HNSW CASES OK" in out.stdout
