"""The global-shutter barrier on the "nccl" backend (RCCL) with a world of one rank: the one-GPU box cannot run two ranks
on the card, but the code path of the barrier - all-reduce(MIN) of an int32 device tensor, pinned copy behind it, event -
is the one every rank of a multi-GPU run executes.  Runs in a process of its own so that the process group does not
leak into the other tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import importlib, os, sys
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from __graft_entry__ import load_pkg
load_pkg()
streams = importlib.import_module("ppg_slam_amd.streams")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sh = streams.ShutterBarrier(dist, "cuda", depth=4, single_rank_too=True)
assert sh.active
log = []
n = streams.run_steps(lambda k: log.append(("s", k)), lambda k: log.append(("w", k)), sh, 0, 25, depth=4)
assert n == 25 and sh.count == 25
assert sh.drain() == 24 and not sh.pending
n = streams.run_steps(lambda k: None, lambda k: None, sh, 25, 3, depth=4)
assert sh.drain() == 27 and sh.count == 28
dist.barrier()
dist.destroy_process_group()
print("rccl barrier ok")
""" % ROOT


@pytest.mark.gpu
def test_shutter_barrier_on_rccl_single_rank():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=110, env=env)
    assert r.returncode == 0 and "rccl barrier ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
