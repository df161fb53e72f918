"""ON THE GPU BOX: which points of the bf16 MLP backward differ between the default build and a variant library?
usage: python tools/dev/k4_diff.py <variant.so> [N]   (runs itself twice as a child process, once per library)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch, ref_cpu
    from hbr_amd import ops
    from hbr_amd._lib import PLANAR, BF16
    N, out = int(sys.argv[2]), sys.argv[3]
    g = torch.Generator().manual_seed(5)
    feat = (torch.randn((16, N, 2), generator=g) * 0.3).to("cuda:0")
    pe = ops.dir_encode(torch.nn.functional.normalize(torch.randn((N, 3), generator=g), dim=1).to("cuda:0"), 4)
    dout = torch.randn((N, 4), generator=g).to("cuda:0")
    P = torch.cat([v.reshape(-1) for v in ref_cpu.mlp_init(2).values()]).to("cuda:0")
    res = {}
    for rep in range(2):
        dP = torch.zeros_like(P)
        df = ops.mlp_bwd(feat, PLANAR, pe, 1, P, BF16, dout, dP)
        torch.cuda.synchronize()
        res[f"df{rep}"] = df.cpu().numpy(); res[f"dP{rep}"] = dP.cpu().numpy()
    np.savez(out, **res)
    sys.exit(0)
var, N = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1000
outs = []
for name, lib in (("default", None), ("variant", os.path.abspath(var))):
    env = dict(os.environ)
    env.pop("HBR_LIB", None)
    if lib: env["HBR_LIB"] = lib
    out = f"/tmp/k4diff_{name}.npz"
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(N), out], env=env, check=True)
    outs.append(np.load(out))
d, v = outs
print("default repeatable:", np.array_equal(d["df0"], d["df1"]), " variant repeatable:", np.array_equal(v["df0"], v["df1"]))
for rep in range(2):
    bad = np.argwhere((d["df0"] != v[f"df{rep}"]).any(axis=(0, 2))).ravel()
    tiles = sorted(set(bad // 32))
    print(f"rep {rep}: {len(bad)} of {N} points differ; tiles {tiles[:40]}{' ...' if len(tiles) > 40 else ''}")
    if len(bad):
        n = bad[0]
        print("  first bad point", n, "default", d["df0"][:4, n].ravel(), "variant", v[f"df{rep}"][:4, n].ravel())
        lv = np.argwhere((d["df0"] != v[f"df{rep}"]).any(axis=(1, 2))).ravel()
        print("  levels touched:", lv, " points within the first bad tile:", bad[bad // 32 == tiles[0]] % 32)
    print(f"  dP max |diff| {np.abs(d['dP0'] - v[f'dP{rep}']).max():.3e} of max {np.abs(d['dP0']).max():.3e}")
