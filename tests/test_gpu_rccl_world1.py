"""GPU (-m gpu): the step's collectives on the REAL backend - RCCL ("nccl") - on the one GPU a test box has.

RCCL refuses two ranks on one device, so the world is ONE rank: an all-reduce then leaves the data as it is, but it
is enqueued on RCCL's own stream and ordered against the compute stream exactly as in the 8-GPU run - which is what the
gloo rehearsal (tests/test_gpu_two_ranks.py; gloo's wait() blocks the host) cannot exercise:
  * StagedAllReduce.launch must order each piece BEHIND the kernel that produced it (K4, then each K2 half);
  * StagedAllReduce.finish must make the optimiser kernel wait for all three pieces.
`trainer.always_reduce` issues the collectives although world == 1.  Result must be bit-identical to the plain step: the
scatter kernel is deterministic (64-bit fixed point) and a one-rank sum changes nothing - any difference is a missing
stream dependency.  Runs in a child process so that the test session's process-group state stays untouched."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from test_gpu_two_ranks import _setup, S
from hbr_amd.trainer import HashNeRFTrainer
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
res = {}
for name, split, always in (("plain", False, False), ("staged", True, True), ("single", False, True)):
    batch, enc, mlp = _setup(dev)
    tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=100, seed=9, split_scatter=split)
    tr.always_reduce = always
    tr.timers = {} if always else None
    losses = [float(tr.step(*batch)) for _ in range(6)]
    torch.cuda.synchronize()
    if always:
        assert "allreduce_exposed" in tr.timers and len(tr.timers["allreduce_exposed"]) == 6, list(tr.timers)
    res[name] = (tr.grad.clone(), tr.tables.clone(), tr.flat.clone(), losses)
for name in ("staged", "single"):
    for a, b, what in zip(res["plain"][:3], res[name][:3], ("grad", "tables", "mlp")):
        assert torch.equal(a, b), (name, what, float((a - b).abs().max()))
    assert res["plain"][3] == res[name][3], (name, res["plain"][3], res[name][3])
# autotune_comm: measures both ways with the collectives issued, keeps one, and puts parameters, moments and the step
# counter back (ADVICE r3): training then goes on exactly like a trainer that never tuned
batch, enc, mlp = _setup(dev)
tr = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=100, seed=9)
tr.always_reduce = True
tune = tr.autotune_comm(lambda i: batch, steps=3)
assert tune["chosen"] in ("single", "staged") and tr.split_scatter == (tune["chosen"] == "staged"), tune
assert tune["single_ms_per_step"] > 0 and tune["staged_ms_per_step"] > 0, tune
after = [float(tr.step(*batch)) for _ in range(2)]
batch, enc, mlp = _setup(dev)
ref = HashNeRFTrainer(enc, mlp, num_samples=S, total_steps=100, seed=9)
ref_losses = [float(ref.step(*batch)) for _ in range(2)]
assert after == ref_losses, (after, ref_losses)
assert tr.step_count == ref.step_count == 2
assert torch.equal(tr.tables, ref.tables) and torch.equal(tr.flat, ref.flat)
torch.distributed.destroy_process_group()
print("RCCL_WORLD1_OK", res["plain"][3][-1], tune)
"""


def test_rccl_one_rank_staged_allreduce_is_ordered_with_the_kernels():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
