"""A few seconds of each random soak (tools/gpu_soak_*.py: random sizes, seeds, modes, launch shapes and API interleavings, GPU
against the CPU oracle - bit-exact for the filters, 1e-7 m + identical LM counts for the pose graph).  The long runs of the
round (minutes each) found four defects the fixed-configuration tests had not; this keeps the harness
alive on a fixed seed (new seeds belong to the tools, not to a gate)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,args", [("gpu_soak_ekf.py", ["both"]), ("gpu_soak_pgs.py", []), ("gpu_soak_api.py", []), ("gpu_soak_adversarial.py", ["both"]), ("gpu_soak_pgs_api.py", [])])
def test_a_few_seconds_of_random_configurations(tool, args):
    seed = "12345"
    cmd = [sys.executable, os.path.join(ROOT, "tools", tool), "8", seed] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])
    assert "0 mismatches" in out.stdout
