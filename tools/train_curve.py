"""PSNR-vs-step curve of the fused trainer on the synthetic lego-shaped scene at the BASELINE size (bf16 vs fp32 MLP)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch, ref_cpu
from hbr_amd._lib import BF16, F32
from hbr_amd.helper import calc_psnr
from hbr_amd.trainer import HashNeRFTrainer, build_default_model
dev = "cuda:0"
R, S, steps = 16000, 128, 1500
o0, d0, _, _ = ref_cpu.synthetic_rays(65536, seed=0)
mn, mx, sig = ref_cpu.bbox_mu_sigma(o0, d0)
pool = [ref_cpu.synthetic_scene_rays(R, seed=100 + i, device=dev) for i in range(64)]
test = ref_cpu.synthetic_scene_rays(R, seed=999, device=dev)
for prec, fdt, name in ((BF16, BF16, "bf16 MLP, bf16 feature buffers"), (BF16, F32, "bf16 MLP, fp32 feature buffers"), (F32, F32, "fp32")):
    enc, denc, mlp = build_default_model(mn, sig, dev, seed=0)
    tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=steps, precision=prec, feat_dtype=fdt)
    torch.manual_seed(0)
    t0 = time.time(); line = [name]
    for k in range(steps + 1):
        if k % 250 == 0:
            torch.manual_seed(12345)
            p = float(calc_psnr(tr.render(test[0], test[1], test[2]), test[3]))
            torch.manual_seed(k + 1)
            line.append(f"step {k}: {p:.2f} dB")
        if k < steps:
            tr.step(*pool[k % len(pool)])
    torch.cuda.synchronize()
    print("; ".join(line), f"; {time.time() - t0:.1f}s", flush=True)
