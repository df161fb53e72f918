"""Dev check of the K2 scatter kernel against a reference golden (prints where the two differ)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ref_cpu
from hbr_amd import ops
from hbr_amd._lib import ROWS
name = sys.argv[1] if len(sys.argv) > 1 else "g3_encoder_T16.npz"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", name)))
L, T = int(g["L"]), int(g["T"])
sc = ref_cpu.level_scales(16, float(g["N_max"]), L)
geom = ops.HashGeom(tuple(float(v) for v in sc), tuple(float(v) for v in g["mu"]), float(g["sigma"]), T, 2)
ref = np.zeros((L, T, 2), np.float32)
if "dtables" in g: ref = g["dtables"]
else: ref[g["dtab_l"], g["dtab_row"]] = g["dtab_val"]
x, dy = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["dy"]).cuda()
print("N", x.shape[0], "T", T, "dy absmax per level", np.abs(g["dy"]).reshape(-1, L, 2).max(axis=(0, 2)))
for algo in (1, 2):
    dt = torch.zeros(ref.shape, device="cuda")
    ops.hash_encode_bwd(geom, dy, dt, x=x, layout=ROWS, algo=algo)
    got = dt.cpu().numpy()
    err = np.abs(got - ref)
    tol = 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max()
    bad = np.argwhere(err > tol)
    print("algo", algo, "max err", err.max(), "bad", len(bad), "touched equal", np.array_equal(got != 0, ref != 0))
    for b in bad[:10]:
        print("  ", tuple(b), got[tuple(b)], ref[tuple(b)])
    per_level = [(l, float(err[l].max())) for l in range(L)]
    print("   per-level max err", ["%d:%.2e" % p for p in per_level])
