"""Race hunt kept in the suite: thousands of dynamic steps with moving thresholds, the pipelined schedule against the
single-graph schedule on identical inputs (tools/stress_schedules.py) - every step bit-identical, LSTM state included."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("steps,n_envs", [(1500, 1), (800, 3)])
def test_pipelined_and_single_graph_schedules_never_diverge(steps, n_envs):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_schedules.py"), "tiny", str(steps), str(n_envs)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout
