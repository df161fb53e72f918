"""Probe: does K1 (texture-bound hash lookup) of one half-batch overlap K4 / K2 (MFMA / VALU+LDS-bound) of the other
half when they are enqueued on two HIP streams?  Prints the serial sum and the concurrent wall time of each pair."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hbr_amd import ops, synthetic
from hbr_amd._lib import BF16, PLANAR
from hbr_amd.trainer import build_default_model
dev = torch.device("cuda", 0)
R, S = 8000, 128
N = R * S
o0, d0, _, _ = synthetic.hemisphere_rays(65536, seed=0)
mn, mx, sig = synthetic.ray_bbox(o0, d0)
enc, _, mlp = build_default_model(mn, sig, dev, seed=0)
geom, tables = enc.geometry(), enc.stacked_tables()
flat, _ = mlp.flat_params()
halves = []
for h in range(2):
    o, d, dn, gt = synthetic.scene_rays(R, seed=10 + h, device=dev)
    t = ops.strat_sample(2.0, 6.0, S, dev, seed=1, offset=h)
    pe = ops.dir_encode(d, 4)
    feat = ops.hash_encode_fwd(geom, tables, rays=(o, d, t), layout=PLANAR, dtype=BF16)
    dout = torch.randn((N, 4), device=dev) * 1e-4
    halves.append(dict(rays=(o, d, t), pe=pe, feat=feat, dout=dout))
# HBR_PROBE_PRIO=1: the K1 stream gets the higher priority (its workgroups are taken first whenever a CU has room)
sa = torch.cuda.Stream()
sb = torch.cuda.Stream(priority=-1) if os.environ.get("HBR_PROBE_PRIO") else torch.cuda.Stream()
g_mlp = torch.zeros_like(flat)
g_tab = torch.zeros_like(tables)
amax = torch.zeros(16, device=dev)


def k1(h): return ops.hash_encode_fwd(geom, tables, rays=halves[h]["rays"], layout=PLANAR, dtype=BF16)
def k4(h): return ops.mlp_bwd(halves[h]["feat"], PLANAR, halves[h]["pe"], S, flat, BF16, halves[h]["dout"], g_mlp, absmax_out=amax)
def k2(h, dfeat): return ops.hash_encode_bwd(geom, dfeat, g_tab, rays=halves[h]["rays"], layout=PLANAR, algo=2)


def timed(fn, reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dfeat = k4(0)
for name, heavy in (("K4", lambda: k4(0)), ("K2", lambda: k2(0, dfeat))):
    for f in (heavy, lambda: k1(1)):
        f()
    t_heavy, t_k1 = timed(heavy), timed(lambda: k1(1))

    def both():
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(sa):
            sa.wait_event(ev)
            heavy()
            ea = torch.cuda.Event(); ea.record()
        with torch.cuda.stream(sb):
            sb.wait_event(ev)
            k1(1)
            eb = torch.cuda.Event(); eb.record()
        torch.cuda.current_stream().wait_event(ea)
        torch.cuda.current_stream().wait_event(eb)
    both()
    t_both = timed(both)
    print(f"{name} alone {t_heavy:.3f} ms, K1 alone {t_k1:.3f} ms, serial sum {t_heavy + t_k1:.3f} ms, two streams {t_both:.3f} ms", flush=True)
