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


def _run_noalloc(ctx, torch, A, B, mode, th, ratio, stream=None):
    nA, nB = A.shape[0], B.shape[0]
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    dm = torch.full((max(nA, 1),), -7, dtype=torch.int32, device="cuda")
    db = torch.full((max(nA, 1),), -7.0, dtype=torch.float32, device="cuda")
    ds = torch.full((max(nA, 1),), -7.0, dtype=torch.float32, device="cuda")
    dc = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = ctx.match_nn_noalloc(dA.data_ptr(), nA, dB.data_ptr(), nB, dm.data_ptr(), db.data_ptr(), ds.data_ptr(), dc.data_ptr(),
                              th=th, ratio=ratio, mode=mode, stream=stream)
    assert rc == 0, ctx.last_error()
    torch.cuda.synchronize()
    return dm.cpu().numpy()[:nA], db.cpu().numpy()[:nA], ds.cpu().numpy()[:nA], int(dc.cpu()[0])


def test_noalloc_match_equals_allocating_entry_point(pkg):
    import torch
    ctx = pkg.Context(pkg.default_config(128, 96))
    try:
        rng = np.random.default_rng(77)
        # two calls in a row per mode, the second smaller; then ragged and degenerate shapes
        for mode in (pkg.MATCH_INIT, pkg.MATCH_PLAIN):
            for (nA, nB) in ((1024, 1024), (300, 217), (1, 1), (130, 0), (64, 1000)):
                A, B = _pair(rng, nA, max(nB, 1))
                B = B[:nB]
                m_ref, b_ref, s_ref, c_ref = ctx.match_nn(A, B, th=0.7, ratio=0.9, mode=mode)
                m, b, s, cnt = _run_noalloc(ctx, torch, A, B, mode, 0.7, 0.9)
                assert np.array_equal(m, m_ref), (mode, nA, nB)
                assert np.array_equal(b, b_ref) and np.array_equal(s, s_ref), (mode, nA, nB)
                if mode == pkg.MATCH_INIT:
                    assert cnt == c_ref, (mode, nA, nB)
                    if nA >= 300 and nB >= 200:   # the claim and ratio logic did fire on real matches
                        assert cnt > min(nA, nB) // 8
        # on a caller's stream the call only enqueues
        A, B = _pair(rng, 400, 380)
        st = torch.cuda.Stream()
        m_ref, b_ref, s_ref, c_ref = ctx.match_nn(A, B, th=0.7, ratio=0.9, mode=pkg.MATCH_INIT)
        m, b, s, cnt = _run_noalloc(ctx, torch, A, B, pkg.MATCH_INIT, 0.7, 0.9, stream=st.cuda_stream)
        assert np.array_equal(m, m_ref) and np.array_equal(b, b_ref) and np.array_equal(s, s_ref) and cnt == c_ref
        # beyond the fixed work memory: a clear error, nothing launched
        big = torch.zeros((1025, 256), dtype=torch.float32, device="cuda")
        out = torch.zeros((1025,), dtype=torch.int32, device="cuda")
        rc = ctx.match_nn_noalloc(big.data_ptr(), 1025, big.data_ptr(), 8, out.data_ptr(), out.data_ptr(), out.data_ptr(),
                                  out.data_ptr(), mode=pkg.MATCH_PLAIN)
        assert rc == pkg.PPG_ECAP and "exceed the fixed work memory" in ctx.last_error()
    finally:
        ctx.close()
