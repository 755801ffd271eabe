"""Concurrency the reference has: one extractor on the tracking thread and Matcher objects built on the stack in the
tracking, local-mapping and loop-closing threads at once (system/src/Tracking.cpp:524-525, 1007;
system/src/LocalMapping.cpp:214-245; system/src/LoopClosing.cpp:408-409).  Here every thread owns a context; the matching
threads own matcher-only ones (ppg_create_matcher).  Results must equal the single-threaded ones bit for bit.
ctypes releases the GIL for the duration of a foreign call, so the calls below do overlap."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 320, 240


def _frames(synth, n):
    scene = synth.make_scene(H, W, 4242, "dense")
    return [synth.frame_from_scene(scene, H, W, t) for t in range(n)]


def test_matcher_context_is_light_and_rejects_the_extractor(pkg, synth):
    full = pkg.Context(pkg.default_config(W, H, max_num=500))
    f0, f1 = full.extract_batch(_frames(synth, 2)[:1])[0], None
    m = pkg.Context(pkg.default_config(W, H, max_num=500), matcher_only=True)
    try:
        # the extractor's entry points say what is wrong instead of touching buffers that do not exist
        with pytest.raises(pkg.PPGError) as e:
            m.extract(_frames(synth, 1)[0])
        assert "ppg_create_matcher" in str(e.value)
        with pytest.raises(pkg.PPGError):
            m.match_frames(0, 0, 1)
        with pytest.raises(pkg.PPGError):
            m.read_tensor(pkg.T_FEAT)
        # every matching entry point works and agrees with the full context
        A = f0["desc"]
        B = np.roll(A, 3, axis=0).copy()
        for mode in (pkg.MATCH_INIT, pkg.MATCH_PLAIN):
            ra = full.match_nn(A, B, th=0.7, ratio=0.9, mode=mode)
            rb = m.match_nn(A, B, th=0.7, ratio=0.9, mode=mode)
            assert all(np.array_equal(x, y) for x, y in zip(ra[:3], rb[:3])) and ra[3] == rb[3]
        assert np.array_equal(full.dist_matrix(A[:40], B[:50]), m.dist_matrix(A[:40], B[:50]))
        assert np.array_equal(full.features_in_area(f0["un"], 100.0, 80.0, 30.0), m.features_in_area(f0["un"], 100.0, 80.0, 30.0))
    finally:
        m.close()
        full.close()


def test_set_tunables_applies_without_a_new_context(pkg, synth, orc, onet):
    """The reference reads its public static tunables on every frame (feature/include/PPGExtractor.h:105-114)."""
    img = _frames(synth, 1)[0]
    cfg = pkg.default_config(W, H, max_num=2000)
    ctx = pkg.Context(cfg)
    try:
        a = ctx.extract(img)
        cfg2 = pkg.default_config(W, H, max_num=2000)
        cfg2.junction_thresh = 0.15
        cfg2.line_heatmap_thresh = 0.3
        assert ctx.set_tunables(cfg2) == 0
        b = ctx.extract(img)
        ocfg = orc.default_cfg(W, H, 2000)
        ocfg.junction_thresh = 0.15
        ocfg.line_heatmap_thresh = 0.3
        o = orc.extract(onet, ocfg, img)
        assert b["n"] == o["n"] < a["n"] and np.array_equal(b["xy"], o["xy"]) and np.array_equal(b["edge_idx"], o["edge_idx"])
        assert ctx.set_tunables(cfg) == 0
        c = ctx.extract(img)
        assert np.array_equal(c["xy"], a["xy"]) and np.array_equal(c["edge_idx"], a["edge_idx"])
        cfg3 = pkg.default_config(W, H, max_num=600)
        assert ctx.set_tunables(cfg3) == pkg.PPG_ECAP and "new context" in ctx.last_error()
    finally:
        ctx.close()


def test_three_threads_extract_and_match_concurrently(pkg, synth):
    imgs = _frames(synth, 6)
    ext = pkg.Context(pkg.default_config(W, H, max_batch=1, max_num=500))
    f0, f1 = ext.extract(imgs[0]), ext.extract(imgs[1])
    rng = np.random.default_rng(5)
    g = synth.random_map_graph(rng, f1["desc"], f1["un"], min(300, f1["n"]), 700)
    m = g["desc"].shape[0]
    uv = f0["un"] + np.array([3.0, 1.0], np.float32)

    def proj(ctx):
        return ctx.match_projected(f1["un"], f1["desc"], f0["desc"], uv, 15.0, 0.8, 0.0, None,
                                   np.ones(f0["n"], np.uint8), np.zeros(f1["n"], np.uint8))

    def extend(ctx):
        return ctx.extend_map_matches(0.8, 5.0, g, f1["desc"], f1["un"], f1["edge_idx"], f1["conn_off"], f1["conn_idx"],
                                      np.zeros(m, np.uint8), np.full(f1["n"], -1, np.int32), np.full(f1["n_edges"], -1, np.int32))

    def init(ctx):
        return ctx.search_for_initialization(0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], 50)

    def same(a, b):
        return all((np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y) for x, y in zip(a, b))

    # single-threaded answers, from the extractor's own context
    want_proj, want_ext, want_init = proj(ext), extend(ext), init(ext)
    assert want_proj[0] > 20 and want_ext[0] > 20 and want_init[0] > 20
    want_frames = [ext.extract(im) for im in imgs]

    errors = []
    start = threading.Barrier(3)

    def tracking():           # the extractor in its pipeline, four batches deep, with the match against the previous frame
        try:
            start.wait()
            for rep in range(6):
                got = []
                done = 0
                for k, im in enumerate(imgs):
                    ext.submit_batch([im], match_prev=False)
                    if k - done + 1 >= 4:
                        got.append(ext.wait_batch()[0][0])
                        done += 1
                while done < len(imgs):
                    got.append(ext.wait_batch()[0][0])
                    done += 1
                for a, b in zip(got, want_frames):
                    if not (np.array_equal(a["xy"], b["xy"]) and np.array_equal(a["edge_idx"], b["edge_idx"]) and np.array_equal(a["desc"], b["desc"])):
                        errors.append("tracking: a pipelined frame differs from the blocking call")
        except Exception as e:  # noqa: BLE001
            errors.append("tracking: %r" % (e,))

    def mapping(name, fns):
        def run():
            ctx = None
            try:
                ctx = pkg.Context(pkg.default_config(W, H, max_num=500), matcher_only=True)
                start.wait()
                for rep in range(12):
                    for fn, want in fns:
                        if not same(fn(ctx), want):
                            errors.append(name + ": a concurrent call differs from the single-threaded one")
            except Exception as e:  # noqa: BLE001
                errors.append(name + ": %r" % (e,))
            finally:
                if ctx is not None:
                    ctx.close()
        return run

    ts = [threading.Thread(target=tracking),
          threading.Thread(target=mapping("local mapping", [(proj, want_proj), (init, want_init)])),
          threading.Thread(target=mapping("loop closing", [(extend, want_ext), (proj, want_proj)]))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=100)
        assert not t.is_alive(), "a thread is stuck"
    ext.close()
    assert not errors, errors[:5]
