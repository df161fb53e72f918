"""Does the ORDER of the rays inside a batch matter to the hash kernels?  K1 / K2 / K4 on the same 16 000 rays x 128 samples
in the loader's random order and sorted by a Morton code of each ray's mid point (neighbouring rays then traverse
neighbouring cells up to level ~10)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hbr_amd import ops, synthetic
from hbr_amd._lib import BF16, PLANAR
from hbr_amd.trainer import build_default_model
dev = torch.device("cuda", 0)
R, S = 16000, 128
o0, d0, _, _ = synthetic.hemisphere_rays(65536, seed=0)
mn, mx, sig = synthetic.ray_bbox(o0, d0, 2.0, 6.0)
enc, _, mlp = build_default_model(mn, sig, dev, seed=0)
geom, tables = enc.geometry(), enc.stacked_tables()
o, d, dn, gt = synthetic.scene_rays(R, seed=1000, device=dev)
t = ops.strat_sample(2.0, 6.0, S, dev, seed=0, offset=0)


def morton(p, bits=10):
    q = ((p - p.min(0).values) / (p.max(0).values - p.min(0).values + 1e-9) * (2 ** bits - 1)).long()
    code = torch.zeros(p.shape[0], dtype=torch.long, device=p.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


mid = o + d * 4.0
orders = {"random": torch.arange(R, device=dev), "morton(mid point)": torch.argsort(morton(mid)),
          "morton(direction, origin)": torch.argsort(morton(torch.cat([d, o], 1)[:, :3]) * 1024 + morton(o, 3))}
dy = (torch.randn((16, R * S, 2), device=dev) * 1e-3).to(torch.bfloat16)
g = torch.zeros_like(tables)
amax = dy.float().abs().amax(dim=(1, 2))
# every order is timed three times, in rotation: the first loop of a process runs on a chip whose clocks and caches are still
# settling, which a single pass in a fixed sequence reads as a difference between the orders
for name, perm in list(orders.items()) * 3:
    oo, dd = o[perm].contiguous(), d[perm].contiguous()
    k1 = timed(lambda: ops.hash_encode_fwd(geom, tables, rays=(oo, dd, t), layout=PLANAR, dtype=BF16))
    k2 = timed(lambda: ops.hash_encode_bwd(geom, dy, g, rays=(oo, dd, t), layout=PLANAR, algo=2, dy_absmax=amax, overwrite=True))
    print(f"{name:28s} K1 {k1:7.1f} us   K2 {k2:7.1f} us", flush=True)
