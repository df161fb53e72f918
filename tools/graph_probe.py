"""Does a hipGraph replay of the fused training step run faster than the eager launches?  Captures HashNeRFTrainer.step
(13 launches) with torch.cuda.graph on fixed inputs - learning rate and step counter frozen at capture time, so this
is a timing probe only, not a trainer - and times replays against eager steps on the same batch."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hbr_amd import synthetic
from hbr_amd.trainer import HashNeRFTrainer, build_default_model
dev = torch.device("cuda", 0)
R, S = 16000, 128
o0, d0, _, _ = synthetic.hemisphere_rays(65536, seed=0)
mn, mx, sig = synthetic.ray_bbox(o0, d0, 2.0, 6.0)
o, d, dn, gt = synthetic.scene_rays(R, seed=1000, device=dev)
batch = tuple(a.contiguous() for a in (o, d, dn.reshape(-1), gt))
enc, denc, mlp = build_default_model(mn, sig, dev, seed=0)
tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=4000000)


def timed(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timed(lambda: tr.step(*batch))
# host time to ENQUEUE one step (no sync): how far ahead of the GPU the launches run
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): tr.step(*batch)
enq = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): tr.step(*batch)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    tr.step(*batch)
replay = timed(g.replay)
print(f"eager step {eager:.4f} ms (host enqueue {enq:.4f} ms/step)   graph replay {replay:.4f} ms", flush=True)
