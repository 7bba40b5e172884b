"""Host side of libdmel_hip.so under AddressSanitizer on a fake HIP runtime (tools/asan/): packers, re-tilers, table builders, workspace
planners, argument checks and launch assembly of every handle run for real, the kernels do not.  GPU sanitizers are not available on
the pool; this is the CPU-build sanitizer run SURVEY.md section 5 lists."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not found")
def test_host_side_is_asan_clean(tmp_path):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan", "run.sh"), str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert "no sanitizer report" in r.stdout
