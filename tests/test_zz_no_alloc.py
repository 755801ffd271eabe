"""GPU test of ppg_match_nn_noalloc, the entry point that allocates nothing: its work memory is a fixed device array
inside the library.  Called through the C ABI with device pointers and compared bit for bit with ppg_match_nn (which
takes host pointers and allocates its work memory), on two calls in a row, the second with a smaller input — the
work array then still holds the first call's distances and lists, which must not leak into the result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit_rows(rng, n):
    a = rng.normal(size=(n, 256)).astype(np.float32)
    return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)


def _pair(rng, nA, nB):
    A = _unit_rows(rng, nA)
    src = rng.integers(0, nA, nB)
    B = A[src] + 0.3 * rng.normal(size=(nB, 256)).astype(np.float32) / 16
    B = (B / np.linalg.norm(B, axis=1, keepdims=True)).astype(np.float32)
    if nB > 7:
        B[5] = B[6]       # an exact tie: the first candidate in order wins
    return A, B


class _Hip:
    """The few HIP runtime calls the test needs for the caller's side (device buffers, a stream), through the runtime
    libppgfront itself is linked against: one HIP runtime in the process, no second one brought in by another package."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipStreamDestroy.argtypes = [C.c_void_p]
        self.live = []

    def _ok(self, rc, what):
        assert rc == 0, f"{what} failed with HIP error {rc}"

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.C.c_void_p()
        self._ok(self.rt.hipMalloc(self.C.byref(p), max(arr.nbytes, 16)), "hipMalloc")
        self.live.append(p)
        if arr.nbytes:
            self._ok(self.rt.hipMemcpy(p, arr.ctypes.data_as(self.C.c_void_p), arr.nbytes, 1), "hipMemcpy H2D")
        return p

    def to_host(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        if out.nbytes:
            self._ok(self.rt.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), p, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def stream(self):
        s = self.C.c_void_p()
        self._ok(self.rt.hipStreamCreate(self.C.byref(s)), "hipStreamCreate")
        return s

    def sync(self):
        self._ok(self.rt.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def free_all(self):
        for p in self.live:
            self.rt.hipFree(p)
        self.live = []


def _run_noalloc(ctx, hip, A, B, mode, th, ratio, stream=None):
    nA, nB = A.shape[0], B.shape[0]
    dA, dB = hip.to_device(A), hip.to_device(B)
    dm = hip.to_device(np.full(max(nA, 1), -7, np.int32))
    db = hip.to_device(np.full(max(nA, 1), -7.0, np.float32))
    ds = hip.to_device(np.full(max(nA, 1), -7.0, np.float32))
    dc = hip.to_device(np.full(1, -7, np.int32))
    hip.sync()
    rc = ctx.match_nn_noalloc(dA.value, nA, dB.value, nB, dm.value, db.value, ds.value, dc.value,
                              th=th, ratio=ratio, mode=mode, stream=stream)
    assert rc == 0, ctx.last_error()
    hip.sync()
    out = (hip.to_host(dm, max(nA, 1), np.int32)[:nA], hip.to_host(db, max(nA, 1), np.float32)[:nA],
           hip.to_host(ds, max(nA, 1), np.float32)[:nA], int(hip.to_host(dc, 1, np.int32)[0]))
    hip.free_all()
    return out


def test_noalloc_match_equals_allocating_entry_point(pkg):
    ctx = pkg.Context(pkg.default_config(128, 96))
    hip = _Hip()
    try:
        rng = np.random.default_rng(77)
        # two calls in a row per mode, the second smaller; then ragged and degenerate shapes
        for mode in (pkg.MATCH_INIT, pkg.MATCH_PLAIN):
            for (nA, nB) in ((1024, 1024), (300, 217), (1, 1), (130, 0), (64, 1000)):
                A, B = _pair(rng, nA, max(nB, 1))
                B = B[:nB]
                m_ref, b_ref, s_ref, c_ref = ctx.match_nn(A, B, th=0.7, ratio=0.9, mode=mode)
                m, b, s, cnt = _run_noalloc(ctx, hip, A, B, mode, 0.7, 0.9)
                assert np.array_equal(m, m_ref), (mode, nA, nB)
                assert np.array_equal(b, b_ref) and np.array_equal(s, s_ref), (mode, nA, nB)
                if mode == pkg.MATCH_INIT:
                    assert cnt == c_ref, (mode, nA, nB)
                    if nA >= 300 and nB >= 200:   # the claim and ratio logic did fire on real matches
                        assert cnt > min(nA, nB) // 8
        # on a caller's stream the call only enqueues
        A, B = _pair(rng, 400, 380)
        st = hip.stream()
        m_ref, b_ref, s_ref, c_ref = ctx.match_nn(A, B, th=0.7, ratio=0.9, mode=pkg.MATCH_INIT)
        m, b, s, cnt = _run_noalloc(ctx, hip, A, B, pkg.MATCH_INIT, 0.7, 0.9, stream=st.value)
        hip.rt.hipStreamDestroy(st)
        assert np.array_equal(m, m_ref) and np.array_equal(b, b_ref) and np.array_equal(s, s_ref) and cnt == c_ref
        # beyond the fixed work memory: a clear error, nothing launched
        big = hip.to_device(np.zeros((1025, 256), np.float32))
        out = hip.to_device(np.zeros(1025, np.int32))
        rc = ctx.match_nn_noalloc(big.value, 1025, big.value, 8, out.value, out.value, out.value, out.value, mode=pkg.MATCH_PLAIN)
        assert rc == pkg.PPG_ECAP and "exceed the fixed work memory" in ctx.last_error()
    finally:
        hip.free_all()
        ctx.close()


def test_noalloc_calls_from_two_contexts_and_threads_do_not_corrupt_each_other(pkg):
    """Every call on a GPU shares one work array.  Two contexts on two host threads, each enqueueing on a stream of its
    own, used to race on it; the library now orders them (a per-GPU lock on the host, an event between streams)."""
    import threading
    ctxs = [pkg.Context(pkg.default_config(128, 96), matcher_only=True) for _ in range(2)]
    hip = _Hip()
    try:
        rng = np.random.default_rng(99)
        jobs = []
        for k, (nA, nB) in enumerate(((1024, 1024), (640, 900))):
            A, B = _pair(rng, nA, nB)
            ref = ctxs[k].match_nn(A, B, th=0.7, ratio=0.9, mode=pkg.MATCH_INIT)
            bufs = dict(dA=hip.to_device(A), dB=hip.to_device(B), dm=hip.to_device(np.zeros(nA, np.int32)),
                        db=hip.to_device(np.zeros(nA, np.float32)), ds=hip.to_device(np.zeros(nA, np.float32)),
                        dc=hip.to_device(np.zeros(1, np.int32)), st=hip.stream())
            jobs.append((nA, nB, ref, bufs))
        hip.sync()
        errors = []
        start = threading.Barrier(2)

        def worker(k):
            nA, nB, ref, b = jobs[k]
            try:
                start.wait()
                for rep in range(25):
                    rc = ctxs[k].match_nn_noalloc(b["dA"].value, nA, b["dB"].value, nB, b["dm"].value, b["db"].value,
                                                  b["ds"].value, b["dc"].value, th=0.7, ratio=0.9, mode=pkg.MATCH_INIT,
                                                  stream=b["st"].value if rep % 5 else None)   # every fifth call blocks on the context's own stream
                    if rc != 0:
                        errors.append((k, rep, ctxs[k].last_error()))
                        return
                    assert hip.rt.hipStreamSynchronize(b["st"]) == 0
                    m = hip.to_host(b["dm"], nA, np.int32)
                    bd = hip.to_host(b["db"], nA, np.float32)
                    cnt = int(hip.to_host(b["dc"], 1, np.int32)[0])
                    if not (np.array_equal(m, ref[0]) and np.array_equal(bd, ref[1]) and cnt == ref[3]):
                        errors.append((k, rep, "result differs from ppg_match_nn"))
            except Exception as e:  # noqa: BLE001
                errors.append((k, repr(e)))

        hip.rt.hipStreamSynchronize.argtypes = [hip.C.c_void_p]
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=100)
            assert not t.is_alive()
        assert not errors, errors[:4]
        for _, _, _, b in jobs:
            hip.rt.hipStreamDestroy(b["st"])
    finally:
        hip.free_all()
        for c in ctxs:
            c.close()
