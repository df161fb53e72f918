"""ms per train step of the DROP-IN route at the BASELINE size: the reference's own loop (route A) -
vol_render under autocast, loss.backward(), torch.optim.Adam / AdamW / CosineAnnealingLR - with the hbr_amd classes,
next to the fused HashNeRFTrainer.step on the same rays."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd.encoder import PositionalEncoder
from hbr_amd.hash_encoding import HashEncoder
from hbr_amd.test_hash import MLP_3D
from hbr_amd.vol_renderer import Volume_Renderer
from hbr_amd.trainer import HashNeRFTrainer, build_default_model
dev = torch.device("cuda", 0)
R, S, steps = 16000, 128, 30
o0, d0, _, _ = ref_cpu.synthetic_rays(65536, seed=0)
mn, mx, sig = ref_cpu.bbox_mu_sigma(o0, d0)
pool = [ref_cpu.synthetic_scene_rays(R, seed=100 + i, device=dev) for i in range(8)]


def timed(fn):
    for i in range(5): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for autocast in (True, False):
    torch.manual_seed(0)
    enc = HashEncoder(N_max=2048.0, N_min=16, L=16, T=2 ** 16, F=2, dim=3, mu=mn.to(dev), sigma=sig.to(dev), device=dev).to(dev)
    nerf = torch.nn.DataParallel(MLP_3D(num_sig=2, num_col=2, L=16, F=2, d_view=24, max_bound=torch.ones(3), min_bound=-torch.ones(3))).to(dev)
    vr = Volume_Renderer(H=800, W=800, K=torch.eye(3), near=2.0, far=6.0, device=dev, Pos_encode=enc,
                         Dir_encode=PositionalEncoder(d_model=3, num_freq=4), max_dim=2 ** 10, sigma_val=sig.to(dev), mu=mn.to(dev))
    oe = torch.optim.Adam(list(enc.Embedding_list.parameters()), lr=0.05)
    om = torch.optim.AdamW(nerf.parameters(), lr=0.005)
    se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=10 ** 6, eta_min=1e-4)
    sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=10 ** 6, eta_min=1e-4)
    crit = torch.nn.MSELoss()

    def step(i):
        o, d, dn, gt = pool[i % 8]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            Cr, Cf, _ = vr.vol_render(nerf, d, o, num_samples=S, update_mask=False, dir_norm=dn, hierarchical=False)
            loss = crit(Cr, gt) + crit(Cf, gt)
        oe.zero_grad(set_to_none=True); om.zero_grad(set_to_none=True)
        loss.backward()
        oe.step(); om.step(); se.step(); sm.step()

    ms = timed(step)
    print(f"drop-in route ({'bf16 autocast' if autocast else 'fp32'}): {ms:.2f} ms/step = {R * S / ms / 1e6:.2f} G ray-samples/s", flush=True)

enc, denc, mlp = build_default_model(mn, sig, dev, seed=0)
tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=10 ** 6)
ms = timed(lambda i: tr.step(*pool[i % 8]))
print(f"fused trainer (bf16): {ms:.2f} ms/step = {R * S / ms / 1e6:.2f} G ray-samples/s", flush=True)
