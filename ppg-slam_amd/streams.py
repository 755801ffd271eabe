"""Stream-parallel sharding of independent camera streams over the GPUs of one node.

Every frame of a stream depends only on that stream's previous frame, so streams are dealt to ranks
(one process per GPU) and never exchange activations.  The only collective is the global-shutter
barrier: an all-reduce(MIN) of each rank's finished frame-group number, so that the results of frame
group k are released on every GPU together.  The message is 4 bytes; it is latency-bound and is the
reason the backend (RCCL over xGMI for "nccl", gloo on CPU) hardly matters.  The reference is a
single-process, single-GPU program (feature/src/PPGExtractor.cpp:41) and has no counterpart.

A stereo rig whose two cameras sit on two GPUs (ranks 2p and 2p+1) may also exchange the frame's
descriptors point to point (StereoExchange: one send and one receive of at most max_num x 256 floats
= 1 MiB, a few microseconds over one xGMI link) so that each side can match left against right with
the library's matcher.

Two implementations of the barrier with one interface (release / drain / count / pending / released):
`Shutter` binds the library's own entry points (ppg_shutter_*: RCCL called directly from C, the path a C++ host uses
and the one bench.py runs on GPUs); `ShutterBarrier` is the same protocol over torch.distributed, which the CPU tests
drive with the gloo backend and bench.py falls back to when the C entry points cannot be set up.
"""
import ctypes as C
import os
import sys

import numpy as np

try:
    import torch
except ImportError:   # the C-ABI binding below does not need it
    torch = None


def _pkg():
    return sys.modules["ppg_slam_amd"]


def _one_rccl_image():
    """The library looks RCCL up at run time and prefers an image that is already mapped.  When the process runs on
    the HIP runtime that a PyTorch wheel bundles, the RCCL built against it is the wheel's too: map that one first."""
    pkg = _pkg()
    m = pkg.mapped_runtimes()
    if m["rccl"]:
        return
    bundled_hip = pkg._torch_bundled("libamdhip64.so")
    if bundled_hip and bundled_hip in m["amdhip64"]:
        p = pkg._torch_bundled("librccl.so")
        if p:
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def shutter_unique_id():
    """ppg_shutter_unique_id: 128 bytes that rank 0 makes and hands to every rank."""
    L = _pkg().lib()
    _one_rccl_image()
    buf = (C.c_uint8 * _pkg().SHUTTER_ID_BYTES)()
    rc = L.ppg_shutter_unique_id(buf)
    if rc != 0:
        raise _pkg().PPGError("ppg_shutter_unique_id failed: " + L.ppg_last_error(None).decode())
    return bytes(buf)


class Shutter:
    """ppg_shutter_*: the global-shutter barrier behind the C ABI (RCCL all-reduce(MIN) on a stream of the library's own,
    result in pinned memory behind an event).  Same interface as ShutterBarrier."""

    def __init__(self, ctx, rank, world, unique_id):
        self.L = _pkg().lib()
        _one_rccl_image()
        self.ctx = ctx
        self.rank, self.world = rank, world
        idbuf = (C.c_uint8 * len(unique_id)).from_buffer_copy(unique_id)
        self.h = self.L.ppg_shutter_create(C.c_void_p(ctx.h), rank, world, idbuf)
        if not self.h:
            raise _pkg().PPGError("ppg_shutter_create failed: " + self.L.ppg_last_error(None).decode())
        self.active = True
        self.released = -1

    def _chk(self, rc):
        if rc != 0:
            raise _pkg().PPGError(f"libppgfront error {rc}: " + self.ctx.last_error())

    def release(self, group: int) -> int:
        r = C.c_int(-1)
        self._chk(self.L.ppg_shutter_release(C.c_void_p(self.h), int(group), C.byref(r)))
        self.released = r.value
        return r.value

    def poll(self) -> int:
        r = C.c_int(-1)
        self._chk(self.L.ppg_shutter_poll(C.c_void_p(self.h), C.byref(r)))
        self.released = r.value
        return r.value

    def drain(self) -> int:
        r = C.c_int(-1)
        self._chk(self.L.ppg_shutter_drain(C.c_void_p(self.h), C.byref(r)))
        self.released = r.value
        return r.value

    def _stats(self):
        a, b = C.c_longlong(0), C.c_int(0)
        self._chk(self.L.ppg_shutter_stats(C.c_void_p(self.h), C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def count(self):
        return self._stats()[0]

    @property
    def pending(self):
        return [None] * self._stats()[1]

    def stereo_swap(self, partner, desc):
        """ppg_stereo_swap: returns the partner's descriptors [m, 256]."""
        desc = np.ascontiguousarray(desc, np.float32).reshape(-1, 256)
        M = int(self.ctx.cfg.junction_max_num)
        out = np.empty((M, 256), np.float32)
        m = C.c_int(0)
        self._chk(self.L.ppg_stereo_swap(C.c_void_p(self.h), int(partner), desc.ctypes.data_as(C.c_void_p), desc.shape[0],
                                         out.ctypes.data_as(C.c_void_p), C.byref(m)))
        return out[:m.value].copy()

    def stereo_match(self, partner, frame=0, nnratio=0.9):
        """ppg_stereo_match: swaps the resident descriptors of `frame` of the last collected batch with `partner` and matches
        this side's rows against the partner's on the GPU.  Returns (matches12[max_num], n_matches, n_partner)."""
        M = int(self.ctx.cfg.junction_max_num)
        m12 = np.full(M, -1, np.int32)
        cnt, npart = C.c_int(0), C.c_int(0)
        self._chk(self.L.ppg_stereo_match(C.c_void_p(self.h), int(partner), int(frame), C.c_float(nnratio),
                                          m12.ctypes.data_as(C.c_void_p), C.byref(cnt), C.byref(npart)))
        return m12, cnt.value, npart.value

    def stereo_match_start(self, partner, frame=0, nnratio=0.9):
        """ppg_stereo_match_start: enqueues swap and match, returns at once."""
        self._chk(self.L.ppg_stereo_match_start(C.c_void_p(self.h), int(partner), int(frame), C.c_float(nnratio)))

    def stereo_match_finish(self):
        """ppg_stereo_match_finish: (matches12[max_num], n_matches, n_partner) of the started match."""
        M = int(self.ctx.cfg.junction_max_num)
        m12 = np.full(M, -1, np.int32)
        cnt, npart = C.c_int(0), C.c_int(0)
        self._chk(self.L.ppg_stereo_match_finish(C.c_void_p(self.h), m12.ctypes.data_as(C.c_void_p), C.byref(cnt), C.byref(npart)))
        return m12, cnt.value, npart.value

    def close(self):
        if getattr(self, "h", None):
            self.L.ppg_shutter_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_streams(n_streams: int, world: int, rank: int):
    """Stream ids owned by `rank`: round-robin, s -> GPU (s mod world)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return list(range(rank, n_streams, world))


class ShutterBarrier:
    """all-reduce(MIN) of the per-rank frame-group counter: group k is released once every rank has finished it.

    release() never blocks on the collective: it starts the all-reduce of this rank's newest finished group (on the
    GPU the result is copied to pinned memory behind it and an event marks its arrival) and returns the newest group
    known to be released everywhere, which may lag by the collectives still in flight.  drain() waits for those.
    Every rank must call release() the same number of times (it is a collective); `count` tells how many were started."""

    def __init__(self, dist, device, depth=8, single_rank_too=False):
        self.dist = dist
        # a world of one needs no barrier; single_rank_too runs the collective anyway (rehearsal of the RCCL path on one GPU)
        self.active = dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_too)
        self.cuda = str(device).startswith("cuda")
        self.depth = depth
        self.count = 0
        self.released = -1
        self.pending = []   # (tensor or pinned copy, handle): oldest first
        if self.active:
            self.dev = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(depth)]
            self.host = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(depth)] if self.cuda else None

    def _done(self, entry):
        buf, handle = entry
        return handle.query() if self.cuda else handle.is_completed()

    def _retire(self, block):
        while self.pending and (block or self._done(self.pending[0])):
            buf, handle = self.pending.pop(0)
            if self.cuda:
                handle.synchronize()
            else:
                handle.wait()
            self.released = max(self.released, int(buf[0]))
            block = False

    def release(self, group: int) -> int:
        self.count += 1
        if not self.active:
            self.released = group
            return group
        if len(self.pending) >= self.depth:
            self._retire(block=True)       # the ring is full: wait for the oldest collective only
        slot = (self.count - 1) % self.depth
        t = self.dev[slot]
        t.fill_(group)
        if self.cuda:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)   # enqueued; the host does not wait
            self.host[slot].copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending.append((self.host[slot], ev))
        else:
            work = self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, async_op=True)
            self.pending.append((t, work))
        self._retire(block=False)
        return self.released

    def drain(self) -> int:
        while self.pending:
            self._retire(block=True)
        return self.released


def run_steps(submit, wait, shutter, first, count, depth=4):
    """`count` steps of a software pipeline `depth` deep: submit(k) enqueues frame group k, wait(k) collects it, and every
    collected group goes through the global-shutter barrier.  Returns the number of groups collected.  The loop of
    bench.py, kept here so that the multi-rank test drives the same code."""
    done = first
    for k in range(first, first + count):
        submit(k)
        if k - done + 1 >= depth:
            wait(done)
            shutter.release(done)
            done += 1
    while done < first + count:
        wait(done)
        shutter.release(done)
        done += 1
    return done - first


class StereoExchange:
    """Point-to-point swap of a frame's descriptors between the two ranks of a stereo pair (2p, 2p+1).

    Buffers are fixed-size ([max_num, dim] plus a count), so the exchange is one send and one receive of
    constant shape per frame and nothing is allocated in steady state."""

    def __init__(self, dist, device, max_num, dim=256):
        self.dist = dist
        self.rank = dist.get_rank()
        world = dist.get_world_size()
        self.partner = self.rank ^ 1
        if self.partner >= world:
            raise ValueError("rank %d has no stereo partner in a world of %d" % (self.rank, world))
        self.max_num, self.dim = max_num, dim
        # row max_num carries the count in its first element
        self.out = torch.zeros(max_num + 1, dim, dtype=torch.float32, device=device)
        self.inp = torch.zeros_like(self.out)

    def swap(self, desc, n):
        """desc: [>= n, dim] float32 tensor on the exchange's device.  Returns (partner's descriptors [m, dim], m)."""
        if n > self.max_num:
            raise ValueError("more descriptors than max_num")
        self.out[:n].copy_(desc[:n])
        self.out[self.max_num, 0] = float(n)
        d = self.dist
        ops = [d.P2POp(d.isend, self.out, self.partner), d.P2POp(d.irecv, self.inp, self.partner)]
        if self.rank & 1:
            ops.reverse()  # the odd rank posts its receive first: send/recv pairs match up without a deadlock
        for r in d.batch_isend_irecv(ops):
            r.wait()
        m = int(self.inp[self.max_num, 0].item())
        return self.inp[:m], m
