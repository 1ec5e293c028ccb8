"""Short runs of the round-4 sweeps (tools/config_fuzz.py, tools/api_fuzz.py, tools/wrapper_fuzz.py) inside the GPU suite: random configurations of
ois_config and random interleavings of the ABI's calls, kernel vs oracle, bit for bit. The long runs are profiles/r4_config_fuzz.txt
(3 000 configurations) and profiles/r4_api_fuzz.txt (36 000 operations)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    res = subprocess.run([sys.executable, os.path.join("tools", script), *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return res.stdout


@pytest.mark.gpu
def test_random_configurations_match_the_oracle():
    out = run("config_fuzz.py", "40", "257", "70")
    assert "40 random configurations" in out and "all bit-identical" in out


@pytest.mark.gpu
def test_random_call_orders_match_the_oracle():
    out = run("api_fuzz.py", "24", "40")
    assert "24 sequences x 40 operations" in out and "all equal to the oracle" in out


@pytest.mark.gpu
def test_random_read_patterns_through_the_python_mirror_match_the_oracle():
    out = run("wrapper_fuzz.py", "20", "90")
    assert "20 sequences x 90 steps" in out and "all equal to the oracle" in out
