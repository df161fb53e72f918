"""`python -m hbr_amd.train_hash2` - the reference's latest trainer (train_hash2.py) on the MI355X path.

Same flags as the reference (train_hash2.py:20-39) plus `--synthetic`/`--steps`/`--precision` for running without a
dataset.  Differences by design: all rays live on the GPU (no DataLoader workers / pinned H2D copies), the step is
`HashNeRFTrainer.step` (explicit kernel pipeline, fused Adam), bf16 needs no GradScaler, and with
`torch.distributed.run` every rank trains on its shard of each batch with one gradient all-reduce per step.
`--hierarchical` takes the autograd route (`Volume_Renderer.vol_render` + torch optimisers), as the reference does.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser(description="Train Hashing (MI355X)")
    p.add_argument("--display", action="store_true", help="(ignored: no GUI)")
    p.add_argument("--compile", action="store_true", help="(ignored: kernels are hand-written HIP)")
    p.add_argument("--load", action="store_true", help="Continue from checkpoint")
    p.add_argument("--update_rate", type=int, default=15, help="Update rate for Occupancy grid (inert, as in the reference)")
    p.add_argument("--write", action="store_true", help="Write images and checkpoints")
    p.add_argument("--num_epochs", type=int, default=1000)
    p.add_argument("--num_batch", type=int, default=16000, help="Ray batch size")
    p.add_argument("--num_imgs", type=int, default=2)
    p.add_argument("--num_samples", type=int, default=64, help="Number of samples along ray")
    p.add_argument("--near", type=float, default=2.0)
    p.add_argument("--far", type=float, default=6.0)
    p.add_argument("--plot_grads", action="store_true", help="(ignored)")
    p.add_argument("--use_sdf", action="store_true", help="(not supported: SDF branch is out of scope)")
    p.add_argument("--hierarchical", action="store_true", help="Use hierarchical sampling")
    p.add_argument("--max_res", type=float, default=2048)
    p.add_argument("--hash_size", type=float, default=16, help="Log size of the hash table")
    p.add_argument("--model_name", type=str, default="default")
    p.add_argument("--data_path", type=str, default=None)
    p.add_argument("--ckpt_name", type=str, default="N_2048_T_16")
    # additions
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic rays instead of a dataset")
    p.add_argument("--steps", type=int, default=0, help="stop after this many optimiser steps (0 = all epochs)")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--out_dir", default="./results")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.use_sdf:
        raise NotImplementedError("--use_sdf: the SDF branch is out of scope")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from . import _lib, checkpoint, dist as hdist
    from .dataset import NeRF_DATA, NeRF_DATA_NEW, intrinsics, materialise_rays
    from .helper import calc_psnr, find_bounding_box2, get_od
    from .trainer import HashNeRFTrainer, build_default_model
    from .vol_renderer import Volume_Renderer

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    dev = torch.device("cuda", torch.cuda.current_device())
    rank, world = hdist.init_from_env(device=dev)
    near, far = args.near, args.far

    # ---- rays (train_hash2.py:50-99) ---------------------------------------------------------------------
    test_rays = None
    if args.synthetic:
        import ref_cpu  # synthetic workload generator only (numpy RNG); not used for compute
        rays_o, rays_d, dir_norms, gts = (a.to(dev) for a in ref_cpu.synthetic_rays(args.synthetic, seed=0))
        test_rays = tuple(a.to(dev) for a in ref_cpu.synthetic_rays(min(args.synthetic, 65536), seed=999))
        H = W = int(np.sqrt(test_rays[0].shape[0]))
    else:
        root = args.data_path if args.data_path is not None else "data/lego/"
        cls = NeRF_DATA if args.data_path is None else NeRF_DATA_NEW
        train_data = cls(json_path=os.path.join(root, "transforms_train.json"))
        test_data = cls(json_path=os.path.join(root, "transforms_tmp.json"))
        K = intrinsics(train_data)
        H, W = int(train_data.H), int(train_data.W)
        rays_o, rays_d, dir_norms, gts = materialise_rays(train_data, K, dev)
        img, c2w, _ = test_data[0]
        o, d, n = get_od(H, W, K.to(dev), c2w[None].to(dev))
        test_rays = (o.reshape(-1, 3), d.reshape(-1, 3), n.reshape(-1, 1), img.permute(1, 2, 0).reshape(-1, 3).to(dev))
    if rank == 0:
        print("SHAPES:", tuple(rays_o.shape), tuple(rays_d.shape), tuple(dir_norms.shape))

    # ---- model (train_hash2.py:106-142) -------------------------------------------------------------------
    L, F, T = 16, 2, int(2 ** args.hash_size)
    max_bound, min_bound = find_bounding_box2([(rays_o, rays_d)], near, far)
    if rank == 0:
        checkpoint.save_bounds(min_bound, max_bound, "bounds_model.npy")
        print("BOUNDING BOX:", max_bound.tolist(), min_bound.tolist())
    sigma = ((max_bound - min_bound) ** 2).sum().sqrt()
    enc, denc, mlp = build_default_model(min_bound, sigma, dev, L=L, F=F, T=T, N_max=args.max_res, seed=0)
    nerf = torch.nn.DataParallel(mlp, device_ids=[dev.index])
    if args.load:
        checkpoint.load_checkpoint(args.ckpt_name, nerf, enc)
    n_batches = max(1, rays_o.shape[0] // args.num_batch)
    total_steps = args.num_epochs * n_batches
    prec = _lib.BF16 if args.precision == "bf16" else _lib.F32
    tr = HashNeRFTrainer(enc, mlp, near=near, far=far, num_samples=args.num_samples, total_steps=total_steps, precision=prec)
    hdist.broadcast_params_([tr.tables, tr.flat])
    vr = Volume_Renderer(H=H, W=W, K=None, near=near, far=far, device=dev, Pos_encode=enc, Dir_encode=denc, max_dim=2 ** 10,
                         sigma_val=sigma, mu=min_bound)
    if args.hierarchical:
        oe = torch.optim.Adam(list(enc.Embedding_list.parameters()), lr=0.05)
        om = torch.optim.AdamW(nerf.parameters(), lr=0.005)
        se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=total_steps, eta_min=1e-4)
        sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=total_steps, eta_min=1e-4)
        crit = torch.nn.MSELoss()

    iters = max(1, n_batches // 100)  # train_hash2.py:189: 100 images per epoch
    step, t0, loss = 0, time.time(), None
    gen = torch.Generator(device=dev).manual_seed(0)  # same shuffle on every rank
    for epoch in range(args.num_epochs):
        perm = torch.randperm(rays_o.shape[0], device=dev, generator=gen)
        for i in range(n_batches):
            idx = perm[i * args.num_batch:(i + 1) * args.num_batch]
            lo, hi = hdist.shard_bounds(idx.shape[0], rank, world)
            idx = idx[lo:hi]
            batch = (rays_o[idx], rays_d[idx], dir_norms[idx], gts[idx])
            if args.hierarchical:
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=prec == _lib.BF16):
                    Cr, Cf, _ = vr.vol_render(nerf, batch[1], batch[0], num_samples=args.num_samples, update_mask=False,
                                              dir_norm=batch[2], hierarchical=True)
                    loss = crit(Cr, batch[3]) + crit(Cf, batch[3])
                loss.backward()
                if world > 1:
                    for p in list(enc.Embedding_list.parameters()) + list(nerf.parameters()):
                        hdist.allreduce_mean_(p.grad, world)
                oe.step(); om.step(); se.step(); sm.step()
                om.zero_grad(set_to_none=True); oe.zero_grad(set_to_none=True)
            else:
                loss = tr.step(*batch)
            step += 1
            if args.write and rank == 0 and step % iters == 0 and test_rays is not None:
                pred = tr.render(test_rays[0], test_rays[1], test_rays[2], num_samples=256)
                os.makedirs(args.out_dir, exist_ok=True)
                if pred.shape[0] == H * W:
                    from PIL import Image
                    im = pred.reshape(H, W, 3)
                    im = ((im - im.min()) / (im.max() - im.min() + 1e-12) * 255).byte().cpu().numpy()  # train_hash2.py:297
                    Image.fromarray(im, "RGB").save(os.path.join(args.out_dir, f"hash_big_diff{epoch}_{i}.png"))
                checkpoint.save_checkpoint(args.model_name, nerf, enc)
                print(f"step {step}: loss {float(loss):.6f} psnr {float(calc_psnr(pred, test_rays[3])):.2f} dB")
            if args.steps and step >= args.steps:
                break
        if rank == 0:
            print(f"Train:{step}:{float(loss):.6f}, Epoch:{epoch}, {step * args.num_batch * args.num_samples / (time.time() - t0):.3e} ray-samples/s")
        if args.steps and step >= args.steps:
            break
    psnr = float(calc_psnr(tr.render(test_rays[0], test_rays[1], test_rays[2], num_samples=args.num_samples), test_rays[3])) if test_rays else float("nan")
    if world > 1:
        torch.distributed.destroy_process_group()
    return {"steps": step, "loss": float(loss), "psnr": psnr}


if __name__ == "__main__":
    print(main())
