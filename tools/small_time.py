"""us per launch of the training step's small kernels at the BASELINE size (16 000 rays x 128 samples): the folded
launches (prologue, compositing + loss + compositing backward, multi-segment Adam) next to the launches they replace."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hbr_amd import ops, synthetic
from hbr_amd._lib import BF16
dev = torch.device("cuda", 0)
R, S = 16000, 128
o, d, dn, gt = synthetic.scene_rays(R, seed=1, device=dev)
dn = dn.reshape(-1).contiguous()
out = torch.randn(R * S, 4, device=dev)
t = ops.strat_sample(2.0, 6.0, S, dev, seed=0, offset=0)
flat = torch.randn(14227, device=dev) * 0.1
n_tab = 16 * 65536 * 2
P, G, M, V = (torch.randn(n_tab + 14228, device=dev) * 1e-3 for _ in range(4))
V.abs_()


def timed(name, fn, n=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:55s} {e0.elapsed_time(e1) / n * 1e3:8.2f} us", flush=True)


def three():
    Cr, _ = ops.composite_fwd(t, out.data_ptr(), 4, out.data_ptr() + 12, 4, dn, R, S, want_wts=False)
    loss, dCr = ops.mse2_loss(Cr, gt)
    d_out = torch.empty_like(out)
    ops.composite_bwd(t, out.data_ptr(), 4, out.data_ptr() + 12, 4, dn, R, S, dCr, d_out.data_ptr(), d_out.data_ptr() + 12)


def sep_pro():
    ops.strat_sample(2.0, 6.0, S, dev, seed=0, offset=1)
    ops.dir_encode(d, 4)
    ops.render_prologue(dev, BF16, params=flat)  # stands in for the pack launch inside mlp_fwd


common = dict(beta1=0.9, beta2=0.999, eps=1e-8, step=7, grad_scale=1.0)
timed("composite_fwd + mse2 + composite_bwd (3 launches + fill)", three)
timed("composite_loss_fwd_bwd (1 launch)", lambda: ops.composite_loss_fwd_bwd(t, out, dn, R, S, gt))
timed("strat + dir_encode + pack (3 launches)", sep_pro)
timed("render_prologue (1 launch)", lambda: ops.render_prologue(dev, BF16, params=flat, rays_d=d, strat=(2.0, 6.0, S, None, 0, 1)))
timed("adam x2", lambda: (ops.adam_step(P[:n_tab], G[:n_tab], M[:n_tab], V[:n_tab], lr=0.05, weight_decay=0.0, **common),
                          ops.adam_step(P[n_tab:n_tab + 14227], G[n_tab:n_tab + 14227], M[n_tab:n_tab + 14227], V[n_tab:n_tab + 14227], lr=0.005, weight_decay=0.01, **common)))
timed("adam_step_multi", lambda: ops.adam_step_multi([dict(p=P[:n_tab], g=G[:n_tab], m=M[:n_tab], v=V[:n_tab], lr=0.05, weight_decay=0.0, **common),
                                                      dict(p=P[n_tab:n_tab + 14227], g=G[n_tab:n_tab + 14227], m=M[n_tab:n_tab + 14227], v=V[n_tab:n_tab + 14227], lr=0.005, weight_decay=0.01, **common)]))
