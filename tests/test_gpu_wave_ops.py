"""GPU (-m gpu): the DPP wave primitives of csrc/wave_reduce.h (64-lane min / max / sum, prefix and suffix sums) in
isolation - two small HIP programs under tools/dev/ compare them with sequential loops on the host.  The kernels that
use them (K2's stripe boxes, K5's transmittance scans) have their own parity tests; this pins the primitives."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["dpp_reduce_test", "dpp_scan_test"])
def test_dpp_wave_primitives(tmp_path, name):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = str(tmp_path / name)
    src = os.path.join(ROOT, "tools", "dev", name + ".hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-o", exe, src], check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr
