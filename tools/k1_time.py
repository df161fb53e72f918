"""Time K1 (hash lookup) alone at the BASELINE size, bf16 planar output; HBR_K1_PAIR=0/1 selects the lane mapping."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hbr_amd import ops, synthetic
from hbr_amd._lib import BF16, F32, PLANAR
from hbr_amd.trainer import build_default_model
dev = torch.device("cuda", 0)
R, S = 16000, 128
o0, d0, _, _ = synthetic.hemisphere_rays(65536, seed=0)
mn, mx, sig = synthetic.ray_bbox(o0, d0, 2.0, 6.0)
enc, _, _ = build_default_model(mn, sig, dev, seed=0)
geom, tables = enc.geometry(), enc.stacked_tables()
o, d, dn, gt = synthetic.scene_rays(R, seed=1000, device=dev)
t = ops.strat_sample(2.0, 6.0, S, dev, seed=0, offset=0)
for dt, name in ((BF16, "bf16"), (F32, "fp32")):
    for _ in range(5): ops.hash_encode_fwd(geom, tables, rays=(o, d, t), layout=PLANAR, dtype=dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.hash_encode_fwd(geom, tables, rays=(o, d, t), layout=PLANAR, dtype=dt)
    e1.record(); torch.cuda.synchronize()
    print(f"HBR_K1_PAIR={os.environ.get('HBR_K1_PAIR', '1')} {name}: hash_fwd {e0.elapsed_time(e1) / 30 * 1e3:.1f} us", flush=True)
