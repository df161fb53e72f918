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


# The reference's flags (train_hash2.py:20-39): names, types and defaults are the contract; the descriptions are ours.
_REFERENCE_FLAGS = (
    ("--display", None, None, "accepted for compatibility; there is no GUI preview"),
    ("--compile", None, None, "accepted for compatibility; the kernels are hand-written HIP, nothing to compile"),
    ("--load", None, None, "start from {ckpt_name}_Nerf_hash.pth / {ckpt_name}_encoder_hash.pth"),
    ("--update_rate", int, 15, "occupancy-grid refresh period (the grid is never refreshed, as in the reference)"),
    ("--write", None, None, "save a test render, both checkpoints and bounds every len(loader)//100 steps"),
    ("--num_epochs", int, 1000, "passes over the ray set"),
    ("--num_batch", int, 16000, "rays per optimisation step (split over the ranks)"),
    ("--num_imgs", int, 2, "unused by this trainer (kept for flag parity)"),
    ("--num_samples", int, 64, "depth samples per ray"),
    ("--near", float, 2.0, "first sample depth"),
    ("--far", float, 6.0, "last stratum's start depth"),
    ("--plot_grads", None, None, "accepted for compatibility; no plotting"),
    ("--use_sdf", None, None, "rejected: the SDF branch is outside the accelerated path"),
    ("--hierarchical", None, None, "add the inverse-CDF second pass (autograd route)"),
    ("--max_res", float, 2048, "finest grid resolution N_max"),
    ("--hash_size", float, 16, "log2 of the rows per level"),
    ("--model_name", str, "default", "prefix of the checkpoints written"),
    ("--data_path", str, None, "scene directory in colmap2nerf format (default: data/lego/ in Blender format)"),
    ("--ckpt_name", str, "N_2048_T_16", "prefix of the checkpoints read by --load"),
)


def build_parser():
    p = argparse.ArgumentParser(description="hash-NeRF trainer on MI355X (flags of the reference's train_hash2.py)")
    for flag, typ, default, text in _REFERENCE_FLAGS:
        if typ is None:
            p.add_argument(flag, action="store_true", help=text)
        else:
            p.add_argument(flag, type=typ, default=default, help=text)
    # additions
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic rays instead of a dataset")
    p.add_argument("--steps", type=int, default=0, help="stop after this many optimiser steps (0 = all epochs)")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--overlap_comm", default="off", choices=["off", "on", "auto"],
                   help="more than one GPU: the step's all-reduce as one collective behind the scatter kernel (off), staged under a "
                        "split scatter launch (on), or measured both ways on the first batches and the faster kept (auto)")
    p.add_argument("--out_dir", default="./results")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.use_sdf:
        raise NotImplementedError("--use_sdf: the SDF branch is out of scope")
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
        from . import synthetic
        rays_o, rays_d, dir_norms, gts = (a.to(dev) for a in synthetic.hemisphere_rays(args.synthetic, seed=0))
        test_rays = tuple(a.to(dev) for a in synthetic.hemisphere_rays(min(args.synthetic, 65536), seed=999))
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
    tr = HashNeRFTrainer(enc, mlp, near=near, far=far, num_samples=args.num_samples, total_steps=total_steps, precision=prec,
                         overlap_comm=args.overlap_comm == "on")
    hdist.broadcast_params_([tr.tables, tr.flat])
    vr = Volume_Renderer(H=H, W=W, K=None, near=near, far=far, device=dev, Pos_encode=enc, Dir_encode=denc, max_dim=2 ** 10,
                         sigma_val=sigma, mu=min_bound)
    if args.hierarchical:
        from .optim import Adam, AdamW  # torch.optim's interface on the fused kernel (one launch per parameter group)
        oe = Adam(list(enc.Embedding_list.parameters()), lr=0.05)
        om = AdamW(nerf.parameters(), lr=0.005)
        se = torch.optim.lr_scheduler.CosineAnnealingLR(oe, T_max=total_steps, eta_min=1e-4)
        sm = torch.optim.lr_scheduler.CosineAnnealingLR(om, T_max=total_steps, eta_min=1e-4)
        crit = torch.nn.MSELoss()

    if world > 1 and args.overlap_comm == "auto" and not args.hierarchical:
        def first_batches(i):  # this rank's shard of the first batches in loader order (ordinary training steps)
            idx = torch.arange(i * args.num_batch, (i + 1) * args.num_batch, device=dev) % rays_o.shape[0]
            lo, hi = hdist.shard_bounds(idx.shape[0], rank, world)
            idx = idx[lo:hi]
            return rays_o[idx], rays_d[idx], dir_norms[idx], gts[idx]
        tune = tr.autotune_comm(first_batches)
        if rank == 0:
            print(f"all-reduce schedule: {tune}")
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
                if world > 1:  # ONE collective over all 28 gradients, as the fused trainer's flat buffer gets
                    hdist.allreduce_mean_grads_([p.grad for p in list(enc.Embedding_list.parameters()) + list(nerf.parameters())], world)
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
                print(f"step {step}: loss {float(loss.detach()):.6f} psnr {float(calc_psnr(pred, test_rays[3])):.2f} dB")
            if args.steps and step >= args.steps:
                break
        if rank == 0:
            print(f"Train:{step}:{float(loss.detach()):.6f}, Epoch:{epoch}, {step * args.num_batch * args.num_samples / (time.time() - t0):.3e} ray-samples/s")
        if args.steps and step >= args.steps:
            break
    psnr = float(calc_psnr(tr.render(test_rays[0], test_rays[1], test_rays[2], num_samples=args.num_samples), test_rays[3])) if test_rays else float("nan")
    if world > 1:
        torch.distributed.destroy_process_group()
    return {"steps": step, "loss": float(loss.detach()), "psnr": psnr}


if __name__ == "__main__":
    print(main())
