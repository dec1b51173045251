"""The opt-in two-wave search kernel (HNSW_MI355X_PAIR=1, hnsw_rs_amd/csrc/pair_kernel.inc): same ids, distance bits,
counts and counters as the CPU oracle.  The switch is read once per process, hence the child process."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT


@pytest.mark.gpu
def test_two_wave_kernel_matches_the_oracle():
    env = dict(os.environ, HNSW_MI355X_PAIR="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "pair_parity.py")], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "PAIR PARITY OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
