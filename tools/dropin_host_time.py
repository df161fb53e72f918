"""Host time to ENQUEUE one step of the reference's loop on the drop-in classes (no synchronisation inside the loop) next to
its GPU time: how close the route is to being host-bound.  cProfile of the enqueue loop with PROFILE=1."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from hbr_amd import synthetic
dev = torch.device("cuda", 0)
o0, d0, _, _ = synthetic.hemisphere_rays(65536, seed=0)
mn, mx, sig = synthetic.ray_bbox(o0, d0, 2.0, 6.0)
batches = []
for b in range(8):
    o, d, dn, gt = synthetic.scene_rays(16000, seed=1000 + b, device=dev)
    batches.append(tuple(a.contiguous() for a in (o, d, dn.reshape(-1), gt)))
import hbr_amd.optim as fused
for name, mod in (("torch.optim", None), ("hbr_amd.optim", fused)):
    # the leg itself (wall per step, synchronised at the ends)
    ms, _ = bench.dropin_leg(dev, batches, mn, sig, 128, 50, 10, 4000000, "bf16", optim=mod)
    # enqueue-only: tiny batches make the GPU work negligible, so the loop's wall time is the host's
    small = [tuple(a[:64].contiguous() for a in b) for b in batches]
    if os.environ.get("PROFILE"):
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        host, _ = bench.dropin_leg(dev, small, mn, sig, 128, 200, 20, 4000000, "bf16", optim=mod)
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    else:
        host, _ = bench.dropin_leg(dev, small, mn, sig, 128, 200, 20, 4000000, "bf16", optim=mod)
    print(f"{name}: {ms:.3f} ms/step at 16000 rays; host-side {host:.3f} ms/step (64-ray batches: GPU work ~0)", flush=True)
