"""CPU: the package's own synthetic workload (hbr_amd/synthetic.py) and the bench launcher.  The package must not
need oracle/ - the oracle keeps an independent copy of the generators for its own tests, and the two must agree so
that a bench run and an oracle run see the same scene."""
import os
import subprocess
import sys

import torch

import ref_cpu
from conftest import ROOT


def test_package_generators_equal_oracle_generators():
    from hbr_amd import synthetic
    a, b = synthetic.hemisphere_rays(513, seed=5), ref_cpu.synthetic_rays(513, seed=5)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    mn, mx, sg = synthetic.ray_bbox(a[0], a[1], 2.0, 6.0)
    mn2, mx2, sg2 = ref_cpu.bbox_mu_sigma(b[0], b[1], 2.0, 6.0)
    assert torch.equal(mn, mn2) and torch.equal(mx, mx2) and torch.equal(sg, sg2)
    s1, s2 = synthetic.scene_rays(300, seed=9, quad=64), ref_cpu.synthetic_scene_rays(300, seed=9, quad=64)
    for x, y in zip(s1, s2):
        assert torch.allclose(x, y, rtol=0, atol=1e-6)
    p = torch.randn(100, 3)
    f1, f2 = synthetic.solid_field(p), ref_cpu.analytic_field(p)
    assert torch.equal(f1[0], f2[0]) and torch.equal(f1[1], f2[1])


def test_package_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "human-body-reconstruction_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".sh")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "ref_cpu" not in text, fn
                assert "oracle" not in text, fn


def test_bench_self_launches_ranks_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE must start its own ranks (before touching the GPU) and hand back
    their exit code.  Without a GPU a rank stops at the 'needs an MI355X' guard (the launcher then tears the other one
    down), which is what we look for, together with the launcher's own failure report naming a child rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.is_available():
        return  # covered by the GPU rehearsal
    assert r.returncode != 0
    log = r.stdout + r.stderr
    assert "bench.py needs an MI355X" in log and "local_rank" in log
