"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI of libppgfront.so.

Three kinds of check:
  * CNN tensors against the golden outputs of the reference's TorchScript nets (tolerance: fp32
    summation-order noise, far inside the 1e-3 descriptor bar);
  * stage-wise: the golden tensors are injected into the context and each post-processing stage must
    reproduce the oracle's integer outputs EXACTLY (same inputs => no rounding excuse);
  * end to end against the oracle / goldens, where near-tied key points may swap (see conftest).
"""
import re

import numpy as np
import pytest

from conftest import golden, assert_keypoints_equivalent, edge_coord_set

pytestmark = pytest.mark.gpu

SMALL = ["cnn_96x128_s1_sparse.npz", "cnn_128x160_s7_dense.npz", "cnn_64x96_s3_sparse.npz"]
FULL = [("full_480x640_s1_sparse.npz", "sparse"), ("full_480x640_s11_dense.npz", "dense"), ("full_512x512_s5_sparse.npz", "sparse")]

_ctx_cache = {}


def get_ctx(pkg, W, H, max_num=500, batch=1, precision=0):
    key = (W, H, max_num, batch, precision)
    if key not in _ctx_cache:
        if len(_ctx_cache) > 6:
            k0 = next(iter(_ctx_cache))
            _ctx_cache.pop(k0).close()
        _ctx_cache[key] = pkg.Context(pkg.default_config(W, H, max_batch=batch, max_num=max_num, precision=precision))
    return _ctx_cache[key]


def nhwc(a):
    return np.ascontiguousarray(np.transpose(a, (1, 2, 0)))


@pytest.mark.parametrize("name", SMALL)
def test_cnn_tensors_vs_reference_nets(name, pkg):
    g = golden(name)
    H, W = g["img"].shape
    ctx = get_ctx(pkg, W, H)
    ctx.inference([g["img"]])
    ctx.detect_keypoints(1)
    ctx.detect_lines(1)  # the last 1x1 conv of EdgeHeatmap + softmax run at the head of this stage
    for which, k, rel in ((pkg.T_FEAT, "feat", 4e-6), (pkg.T_JUNC, "junc", 4e-6), (pkg.T_HEAT_RAW, "heat", 1e-5),
                          (pkg.T_DESCMAP, "desc", 4e-6)):
        got = ctx.read_tensor(which)
        ref = nhwc(g[k])
        err, scale = float(np.abs(got - ref).max()), float(np.abs(ref).max())
        assert err <= rel * scale, (k, err, scale)
    assert float(np.abs(ctx.read_tensor(pkg.T_SCORE) - g["score"]).max()) < 1e-5
    assert float(np.abs(ctx.read_tensor(pkg.T_HEAT) - g["prob"]).max()) < 2e-5


@pytest.mark.parametrize("name", SMALL)
def test_stages_exact_on_injected_tensors(name, pkg, orc):
    g = golden(name)
    H, W = g["img"].shape
    ctx = get_ctx(pkg, W, H)
    ctx.inference([g["img"]])
    ctx.write_tensor(pkg.T_SCORE, g["score"])
    ctx.write_tensor(pkg.T_HEAT, g["prob"])
    ctx.write_tensor(pkg.T_DESCMAP, nhwc(g["desc"]))
    ctx.detect_keypoints(1)
    ctx.detect_lines(1)
    ctx.gen_descriptors(1)
    r = ctx.collect(1)[0]
    assert np.array_equal(r["xy"], g["kp_xy"])
    assert np.array_equal(r["score"], g["kp_score"])
    assert np.array_equal(r["un"], g["kp_un"])
    assert np.array_equal(r["out"], g["kp_out"])
    assert np.array_equal(ctx.read_tensor(pkg.T_HEAT_REFINED), g["refined"])  # bit-exact refineHeatMap
    assert np.array_equal(r["edge_idx"], g["edge_idx"])
    assert np.array_equal(r["edge_score"], g["edge_score"], equal_nan=True)
    assert np.array_equal(r["conn_off"], g["conn_off"]) and np.array_equal(r["conn_idx"], g["conn_idx"])
    assert np.array_equal(r["coline_off"], g["coline_off"]) and np.array_equal(r["coline_pairs"], g["coline_pairs"])
    # K11/K12 vs torch grid_sampler + normalize (required tolerance 1e-3)
    assert float(np.linalg.norm(r["desc"] - g["kp_desc"], axis=1).max()) < 1e-5


@pytest.mark.parametrize("name,kind", FULL)
def test_full_frame_vs_golden(name, kind, pkg, synth):
    g = golden(name)
    H, W, seed, maxnum = [int(v) for v in g["meta"]]
    ctx = get_ctx(pkg, W, H, maxnum)
    for t in (0, 1):
        img = synth.make_frame(H, W, seed, t, kind)
        r = ctx.extract(img)
        assert float(np.abs(ctx.read_tensor(pkg.T_SCORE)[::4, ::4] - g[f"score_sub_{t}"]).max()) < 1e-5
        assert float(np.mean(np.abs(ctx.read_tensor(pkg.T_HEAT_REFINED)[::4, ::4] - g[f"refined_sub_{t}"]) > 1e-4)) < 1e-3
        swaps = assert_keypoints_equivalent(r["xy"], r["score"], g[f"kp_xy_{t}"], g[f"kp_score_{t}"])
        ea, eb = edge_coord_set(r["edge_idx"], r["xy"]), edge_coord_set(g[f"edge_idx_{t}"], g[f"kp_xy_{t}"])
        assert len(ea ^ eb) <= 2, len(ea ^ eb)
        if swaps == 0:
            # descriptors: ||d_gpu - d_ref||_2 <= 1e-3 is the bar; fp32 MFMA lands near 1e-5
            assert float(np.linalg.norm(r["desc"] - g[f"kp_desc_{t}"], axis=1).max()) < 1e-4
            if len(ea ^ eb) == 0:
                assert np.array_equal(r["edge_idx"], g[f"edge_idx_{t}"])


def test_ragged_sizes_vs_oracle(pkg, orc, onet, synth):
    """W not a multiple of the 32-px conv tile, H not of 8 rows x 2; refine tiles with remainders."""
    for (H, W, seed) in ((72, 104, 21), (480, 752, 22)):
        img = synth.make_frame(H, W, seed, 0, "sparse")
        ctx = get_ctx(pkg, W, H)
        r = ctx.extract(img)
        cfg = orc.default_cfg(W, H)
        t = onet.inference(img)
        feat = ctx.read_tensor(pkg.T_FEAT)
        assert float(np.abs(feat - nhwc(t["feat"])).max()) <= 4e-6 * float(np.abs(t["feat"]).max())
        # post-processing exact on the GPU's own tensors
        kp = orc.detect_keypoints(cfg, ctx.read_tensor(pkg.T_SCORE))
        assert np.array_equal(r["xy"], kp["xy"]) and np.array_equal(r["un"], kp["un"])
        refined = orc.refine_heat(cfg, ctx.read_tensor(pkg.T_HEAT))
        assert np.array_equal(ctx.read_tensor(pkg.T_HEAT_REFINED), refined)
        ln = orc.detect_lines(cfg, refined, kp["un"], kp["out"])
        assert np.array_equal(r["edge_idx"], ln["edge_idx"]) and np.array_equal(r["conn_idx"], ln["conn_idx"])
        assert np.array_equal(r["coline_pairs"], ln["coline_pairs"])
        d = orc.gen_descriptors(cfg, np.transpose(ctx.read_tensor(pkg.T_DESCMAP), (2, 0, 1)), kp["xy"])
        assert float(np.linalg.norm(r["desc"] - d, axis=1).max()) < 1e-5


def test_smallest_image_sizes_vs_oracle(pkg, orc, onet):
    """The smallest images ppg_create accepts (sides of 16 and up, multiples of 8), one and three frames per call: a
    feature map of 2 x 2 cells, images narrower than a convolution tile, the 4-pixel border leaving room for no key
    point at all.  Network outputs within the fp32 band of the oracle's, key points, edges, adjacency, collinear pairs
    equal, N = 0 where the oracle finds nothing."""
    rng = np.random.default_rng(5)

    def scene(H, W):
        img = np.full((H, W), 128, np.uint8)
        for _ in range(max(2, H * W // 400)):
            y0, x0 = int(rng.integers(0, H - 3)), int(rng.integers(0, W - 3))
            h, w = int(rng.integers(3, max(4, H // 2))), int(rng.integers(3, max(4, W // 2)))
            img[y0:y0 + h, x0:x0 + w] = int(rng.integers(0, 256))
        return img

    found = 0
    for (H, W) in ((16, 16), (16, 32), (24, 40), (32, 16), (40, 24), (16, 200), (200, 16), (48, 48), (56, 72)):
        for B in (1, 3):
            imgs = [scene(H, W) for _ in range(B)]
            ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=500))
            try:
                outs = ctx.extract_batch(imgs)
                feat = [ctx.read_tensor(pkg.T_FEAT, b) for b in range(B)]
                junc = [ctx.read_tensor(pkg.T_JUNC, b) for b in range(B)]
            finally:
                ctx.close()
            for b in range(B):
                t = onet.inference(imgs[b])
                assert float(np.abs(feat[b] - nhwc(t["feat"])).max()) <= 4e-6 * max(float(np.abs(t["feat"]).max()), 1.0), (H, W, B, b)
                assert float(np.abs(junc[b] - nhwc(t["junc"])).max()) <= 4e-6 * max(float(np.abs(t["junc"]).max()), 1.0), (H, W, B, b)
                o = orc.extract(onet, orc.default_cfg(W, H, 500), imgs[b])
                r = outs[b]
                assert r["n"] == o["n"], (H, W, B, b)
                if o["n"] == 0:
                    continue
                found += r["n"]
                assert np.array_equal(r["xy"], o["xy"]) and np.array_equal(r["out"], o["out"]), (H, W, B, b)
                assert np.array_equal(r["edge_idx"], o["edge_idx"]) and np.array_equal(r["conn_idx"], o["conn_idx"]), (H, W, B, b)
                assert np.array_equal(r["coline_pairs"], o["coline_pairs"]), (H, W, B, b)
                assert float(np.linalg.norm(r["desc"] - o["desc"], axis=1).max()) < 1e-5, (H, W, B, b)
    assert found > 20


def test_end_to_end_vs_oracle_random_scenes(pkg, orc, synth):
    """Whole frames, image in -> key points, edges, adjacency, collinear pairs, descriptors out, against the oracle running
    its own convolutions and its own libm, over a set of random sizes, scenes and JUNCTION_MAX_NUM values.  The two
    evaluations differ by ~1e-6 in the score map (CPU and GPU exponentials), so a frame is compared exactly in its integer
    outputs when no two competing scores are that close (nearly all frames) and by sets otherwise; scores 1e-5, edge
    scores 1e-4, descriptors 1e-5."""
    rng = np.random.default_rng(2024)
    net = orc.Net()
    cases = [(96, 128, "sparse", 500), (64, 96, "dense", 500)]
    for _ in range(8):
        cases.append((int(rng.integers(8, 31)) * 8, int(rng.integers(12, 41)) * 8, ("sparse", "dense")[int(rng.integers(0, 2))],
                      (60, 500, 1024)[int(rng.integers(0, 3))]))
    total_kp = total_edges = exact = 0
    for k, (H, W, kind, maxnum) in enumerate(cases):
        img = synth.make_frame(H, W, 3000 + k, int(rng.integers(0, 3)), kind)
        ctx = pkg.Context(pkg.default_config(W, H, max_num=maxnum))
        try:
            r = ctx.extract(img)
        finally:
            ctx.close()
        o = orc.extract(net, orc.default_cfg(W, H, maxnum), img)
        tag = (H, W, kind, maxnum)
        assert r["n"] == o["n"], tag
        total_kp += r["n"]
        total_edges += r["n_edges"]
        if not np.array_equal(r["xy"], o["xy"]):
            assert_keypoints_equivalent(r["xy"], r["score"], o["xy"], o["score"])
            ea, eb = edge_coord_set(r["edge_idx"], r["xy"]), edge_coord_set(o["edge_idx"], o["xy"])
            assert len(ea ^ eb) <= 2, (tag, len(ea ^ eb))
            continue
        assert float(np.abs(r["score"] - o["score"]).max(initial=0.0)) < 1e-5, tag
        assert np.array_equal(r["un"], o["un"]) and np.array_equal(r["out"], o["out"]), tag
        if not np.array_equal(r["edge_idx"], o["edge_idx"]):   # an edge score within rounding noise of a threshold
            ea, eb = edge_coord_set(r["edge_idx"], r["xy"]), edge_coord_set(o["edge_idx"], o["xy"])
            assert len(ea ^ eb) <= 2, (tag, len(ea ^ eb))
            continue
        assert np.allclose(r["edge_score"], o["edge_score"], rtol=0, atol=1e-4, equal_nan=True), tag
        assert np.array_equal(r["conn_off"], o["conn_off"]) and np.array_equal(r["conn_idx"], o["conn_idx"]), tag
        assert np.array_equal(r["coline_off"], o["coline_off"]) and np.array_equal(r["coline_pairs"], o["coline_pairs"]), tag
        if r["n"]:
            assert float(np.linalg.norm(r["desc"] - o["desc"], axis=1).max()) < 1e-5, tag
        exact += 1
    print(f"end to end: {exact} of {len(cases)} frames equal in every integer output, {total_kp} key points, {total_edges} edges")
    assert exact >= len(cases) - 2 and total_kp > 600 and total_edges > 300


def test_full_size_postprocessing_exact_on_gpu_tensors(pkg, orc, synth):
    """The benchmark image sizes: every integer output equals the oracle's when both start from the GPU's CNN tensors."""
    for (H, W, seed, kind, maxnum) in ((480, 640, 31, "sparse", 500), (480, 640, 32, "dense", 1024), (512, 512, 33, "dense", 1024),
                                       (1024, 1024, 34, "sparse", 1024), (1024, 1024, 35, "dense", 1024)):
        ctx = get_ctx(pkg, W, H, maxnum)
        img = synth.make_frame(H, W, seed, 0, kind)
        r = ctx.extract(img)
        cfg = orc.default_cfg(W, H, maxnum)
        kp = orc.detect_keypoints(cfg, ctx.read_tensor(pkg.T_SCORE))
        assert r["n"] == kp["n"]
        if (H, kind) == (1024, "dense"):
            # more than 8192 NMS survivors: the ranking kernel takes its radix-select path
            full = orc.default_cfg(W, H, 1 << 20)
            assert orc.detect_keypoints(full, ctx.read_tensor(pkg.T_SCORE))["n"] > 8192
        assert np.array_equal(r["xy"], kp["xy"]) and np.array_equal(r["score"], kp["score"])
        refined = orc.refine_heat(cfg, ctx.read_tensor(pkg.T_HEAT))
        assert np.array_equal(ctx.read_tensor(pkg.T_HEAT_REFINED), refined)
        ln = orc.detect_lines(cfg, refined, kp["un"], kp["out"])
        assert np.array_equal(r["edge_idx"], ln["edge_idx"])
        assert np.array_equal(r["edge_score"], ln["edge_score"], equal_nan=True)
        assert np.array_equal(r["conn_off"], ln["conn_off"]) and np.array_equal(r["conn_idx"], ln["conn_idx"])
        assert np.array_equal(r["coline_off"], ln["coline_off"]) and np.array_equal(r["coline_pairs"], ln["coline_pairs"])


def test_refine_variants_vs_oracle(pkg, orc, synth):
    """refineHeatMap with other tile sizes (8: the one-wave-per-tile kernel with quarter-filled lanes; 12: tiles with
    remainders, one work-group per tile) and with a validity threshold below 2^-20, where the top-30 % mean is no longer
    exact in any order and both kernels take their sorted, sequential summation."""
    H, W = 96, 128
    img = synth.make_frame(H, W, 77, 0, "dense")
    for (sz, vth) in ((8, 1.0e-2), (12, 1.0e-2), (16, 1.0e-7), (12, 1.0e-7)):
        gcfg = pkg.default_config(W, H, max_batch=1, max_num=500)
        gcfg.heatmap_refine_sz = sz
        gcfg.line_valid_thresh = vth
        ctx = pkg.Context(gcfg)
        try:
            ctx.extract(img)
            ocfg = orc.default_cfg(W, H)
            ocfg.heatmap_refine_sz = sz
            ocfg.line_valid_thresh = vth
            refined = orc.refine_heat(ocfg, ctx.read_tensor(pkg.T_HEAT))
            got = ctx.read_tensor(pkg.T_HEAT_REFINED)
            assert np.array_equal(got, refined), (sz, vth, int(np.sum(got != refined)))
            assert float(refined.max()) > 0.0
        finally:
            ctx.close()


def test_quarter_tiles_equal_big_tiles(pkg, synth, monkeypatch):
    """Small batches run the fp32 3x3 layers in quarter-size work-groups (4x32 / 8x16 pixels x 32 couts, pooling inside
    the 8x16 tile); PPG_NO_QUARTER_TILES=1 keeps the big tiles.  Every network output must be bit-identical, on even
    and on ragged sizes (the chain of every output is the same in both tilings)."""
    for (H, W, n) in ((480, 640, 1), (96, 128, 3), (72, 104, 2), (120, 136, 1)):
        imgs = [synth.make_frame(H, W, 610 + k, 0, "dense" if k % 2 else "sparse") for k in range(n)]
        small = pkg.Context(pkg.default_config(W, H, max_batch=n, max_num=500))
        monkeypatch.setenv("PPG_NO_QUARTER_TILES", "1")
        big = pkg.Context(pkg.default_config(W, H, max_batch=n, max_num=500))
        monkeypatch.delenv("PPG_NO_QUARTER_TILES")
        try:
            rs, rb = small.extract_batch(imgs), big.extract_batch(imgs)
            for which in (pkg.T_FEAT, pkg.T_JUNC, pkg.T_HEAT_RAW, pkg.T_DESCMAP, pkg.T_SCORE, pkg.T_HEAT):
                for f in range(n):
                    assert np.array_equal(small.read_tensor(which, f), big.read_tensor(which, f)), (H, W, n, which, f)
            for a, b in zip(rs, rb):
                assert np.array_equal(a["xy"], b["xy"]) and np.array_equal(a["desc"], b["desc"])
                assert np.array_equal(a["edge_idx"], b["edge_idx"])
        finally:
            small.close()
            big.close()


def test_xcd_order_equals_plain_order(pkg, synth, monkeypatch):
    """Convolution launches of one work-group per tile are dispatched in an XCD-aware order (bands of tiles per XCD, the
    cout blocks of a group sharing a tile back to back); PPG_XCD_MAP=0 keeps the plain (tile, cout block) grid, other
    values other group sizes, PPG_XCD_BAND=0 the interleaved tile assignment.  The order must not change a bit of any
    output, in any arithmetic mode, on sizes whose tile counts are not multiples of eight, with and without quarter-size
    work-groups."""
    for (H, W, n, prec) in ((480, 640, 1, 0), (96, 128, 3, 0), (72, 104, 2, 3), (120, 136, 5, 1), (480, 640, 6, 0)):
        imgs = [synth.make_frame(H, W, 910 + k, 0, "dense" if k % 2 else "sparse") for k in range(n)]
        ctxs = []
        for env in ({"PPG_XCD_MAP": "0"}, {}, {"PPG_XCD_MAP": "64"}, {"PPG_XCD_MAP": "3", "PPG_XCD_BAND": "0"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            ctxs.append(pkg.Context(pkg.default_config(W, H, max_batch=n, max_num=500, precision=prec)))
            for k in env:
                monkeypatch.delenv(k)
        try:
            res = [c.extract_batch(imgs) for c in ctxs]
            for c, r in zip(ctxs[1:], res[1:]):
                for which in (pkg.T_FEAT, pkg.T_JUNC, pkg.T_HEAT_RAW, pkg.T_DESCMAP, pkg.T_SCORE, pkg.T_HEAT):
                    for f in range(n):
                        assert np.array_equal(ctxs[0].read_tensor(which, f), c.read_tensor(which, f)), (H, W, n, prec, which, f)
                for a, b in zip(res[0], r):
                    assert np.array_equal(a["xy"], b["xy"]) and np.array_equal(a["desc"], b["desc"])
                    assert np.array_equal(a["edge_idx"], b["edge_idx"])
        finally:
            for c in ctxs:
                c.close()


def test_wave_specialised_convolutions_equal_plain_ones(pkg, synth, monkeypatch):
    """The reduced-precision 3x3 layers of large launches run wave-specialised (conv_ws_kernel: one resident work-group per
    CU, producer waves staging the next chunk for consumer waves, tiles drawn from per-XCD counters that the kernel leaves
    at zero: every context below extracts twice); PPG_CONV_WS=0 keeps every layer on the one-tile-per-work-group kernel.
    Same arithmetic in the same order: every network output must be bit-identical, in modes 1, 2 and 3, on sizes whose tile
    counts do and do not divide evenly, with every layer forced onto the new kernel (PPG_CONV_WS_MIN=8) and with the
    production threshold."""
    cases = ((480, 640, 4, 1, "8"), (480, 640, 3, 3, "8"), (120, 136, 5, 1, "8"), (72, 104, 7, 3, "8"), (96, 128, 2, 2, "8"),
             (480, 640, 5, 1, None), (240, 320, 1, 1, "8"), (64, 96, 1, 1, "8"), (200, 328, 2, 3, "8"), (480, 640, 1, 2, "8"),
             (256, 512, 9, 1, None))
    for (H, W, n, prec, ws_min) in cases:
        imgs = [synth.make_frame(H, W, 1310 + k, 0, "dense" if k % 2 else "sparse") for k in range(n)]
        monkeypatch.setenv("PPG_CONV_WS", "0")
        plain = pkg.Context(pkg.default_config(W, H, max_batch=n, max_num=500, precision=prec))
        monkeypatch.setenv("PPG_CONV_WS", "1")
        if ws_min:
            monkeypatch.setenv("PPG_CONV_WS_MIN", ws_min)
        spec = pkg.Context(pkg.default_config(W, H, max_batch=n, max_num=500, precision=prec))
        monkeypatch.delenv("PPG_CONV_WS")
        monkeypatch.delenv("PPG_CONV_WS_MIN", raising=False)
        try:
            for rep in range(2):
                ra, rb = plain.extract_batch(imgs), spec.extract_batch(imgs)
                for which in (pkg.T_FEAT, pkg.T_JUNC, pkg.T_HEAT_RAW, pkg.T_DESCMAP):
                    for f in range(n):
                        assert np.array_equal(plain.read_tensor(which, f), spec.read_tensor(which, f)), (H, W, n, prec, which, f, rep)
                for x, y in zip(ra, rb):
                    assert np.array_equal(x["xy"], y["xy"]) and np.array_equal(x["desc"], y["desc"]) and np.array_equal(x["edge_idx"], y["edge_idx"])
        finally:
            plain.close()
            spec.close()


def test_wave_specialised_convolutions_many_batches_in_a_row(pkg, synth, monkeypatch):
    """The wave-specialised kernel hands tiles, records and LDS stages from wave to wave through barriers and counters: a
    hazard there would show up now and then, not every time.  Forty 16-frame batches in a row (the production threshold, mode
    1, frames changing from batch to batch, the pipelined entry points four deep), every network output of every batch
    bit-identical to the one-tile-per-work-group kernel's."""
    H, W, n, T = 480, 640, 16, 5
    scenes = [synth.make_scene(H, W, 4100 + 7 * k, "dense" if k % 3 else "sparse") for k in range(n)]
    frames = [[synth.frame_from_scene(scenes[k], H, W, t) for k in range(n)] for t in range(T)]
    monkeypatch.setenv("PPG_CONV_WS", "0")
    plain = pkg.Context(pkg.default_config(W, H, max_batch=n, max_num=500, precision=1))
    monkeypatch.setenv("PPG_CONV_WS", "1")
    spec = pkg.Context(pkg.default_config(W, H, max_batch=n, max_num=500, precision=1))
    monkeypatch.delenv("PPG_CONV_WS")
    try:
        want = []
        for t in range(T):
            plain.extract_batch(frames[t], to_dict=False)
            want.append([plain.read_tensor(which, f).copy() for which in (pkg.T_FEAT, pkg.T_DESCMAP) for f in (0, n - 1)])
        for rep in range(40):
            t = rep % T
            spec.extract_batch(frames[t], to_dict=False)
            got = [spec.read_tensor(which, f) for which in (pkg.T_FEAT, pkg.T_DESCMAP) for f in (0, n - 1)]
            for x, y in zip(want[t], got):
                assert np.array_equal(x, y), (rep, t)
        # and through the pipeline, where the convolutions of one batch run beside the post-processing of another
        ref = [plain.extract_batch(frames[t], to_dict=True) for t in range(T)]
        done = 0
        for k in range(24):
            spec.submit_batch(frames[k % T], match_prev=False)
            if k - done + 1 >= 4:
                fr, _, _, cnt = spec.wait_batch(to_dict=True, want_matches=False)
                for x, y in zip(ref[done % T], fr):
                    assert np.array_equal(x["xy"], y["xy"]) and np.array_equal(x["desc"], y["desc"]) and np.array_equal(x["edge_idx"], y["edge_idx"]), done
                done += 1
        while done < 24:
            fr, _, _, cnt = spec.wait_batch(to_dict=True, want_matches=False)
            for x, y in zip(ref[done % T], fr):
                assert np.array_equal(x["xy"], y["xy"]) and np.array_equal(x["desc"], y["desc"]), done
            done += 1
    finally:
        plain.close()
        spec.close()


def test_tile_walk_equals_one_group_per_tile(pkg, synth, monkeypatch):
    """Large launches of the fused first convolution run a fixed number of work-groups that take their tiles from a counter
    (which the kernel leaves at zero for the next launch: every context below extracts twice);
    PPG_CONV_PERSIST="f,o" sets the number for the fused / the other fp32 layers, 0 = one work-group per tile.  Every network output must be bit-identical,
    with tile counts that do and do not divide evenly, on even and on ragged sizes."""
    for (H, W, n, walk) in ((480, 640, 2, "64,0"), (480, 640, 1, "7,5"), (120, 136, 3, "8,3"), (72, 104, 4, "1,1")):
        imgs = [synth.make_frame(H, W, 710 + k, 0, "dense" if k % 2 else "sparse") for k in range(n)]
        monkeypatch.setenv("PPG_NO_QUARTER_TILES", "1")
        monkeypatch.setenv("PPG_CONV_PERSIST", "0,0")
        plain = pkg.Context(pkg.default_config(W, H, max_batch=n, max_num=500))
        monkeypatch.setenv("PPG_CONV_PERSIST", walk)
        walking = pkg.Context(pkg.default_config(W, H, max_batch=n, max_num=500))
        monkeypatch.delenv("PPG_CONV_PERSIST")
        monkeypatch.delenv("PPG_NO_QUARTER_TILES")
        try:
            walking.extract_batch(imgs[::-1])
            rp, rw = plain.extract_batch(imgs), walking.extract_batch(imgs)
            for which in (pkg.T_FEAT, pkg.T_JUNC, pkg.T_HEAT_RAW, pkg.T_DESCMAP, pkg.T_SCORE, pkg.T_HEAT):
                for f in range(n):
                    assert np.array_equal(plain.read_tensor(which, f), walking.read_tensor(which, f)), (H, W, n, walk, which, f)
            for a, b in zip(rp, rw):
                assert np.array_equal(a["xy"], b["xy"]) and np.array_equal(a["desc"], b["desc"])
                assert np.array_equal(a["edge_idx"], b["edge_idx"])
        finally:
            plain.close()
            walking.close()


def test_line_tail_large_candidate_lists(pkg, orc, synth):
    """More line candidates than the relaxation's LDS arrays hold (20480; this frame has 21450): the device tail keeps its
    state in global memory.
    Edges, adjacency and collinearity lists against the oracle on the GPU's own tensors, and the frame must not have gone
    to the host tail."""
    H, W, maxnum = 1024, 1024, 1600
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=1, max_num=maxnum))
    try:
        img = synth.make_frame(H, W, 35, 0, "dense")
        f0, h0 = ctx.tail_stats()
        r = ctx.extract(img)
        f1, h1 = ctx.tail_stats()
        cfg = orc.default_cfg(W, H, maxnum)
        kp = orc.detect_keypoints(cfg, ctx.read_tensor(pkg.T_SCORE))
        assert r["n"] == kp["n"] == maxnum
        refined = orc.refine_heat(cfg, ctx.read_tensor(pkg.T_HEAT))
        ln = orc.detect_lines(cfg, refined, kp["un"], kp["out"])
        assert np.array_equal(r["edge_idx"], ln["edge_idx"])
        assert np.array_equal(r["conn_off"], ln["conn_off"]) and np.array_equal(r["conn_idx"], ln["conn_idx"])
        assert np.array_equal(r["coline_off"], ln["coline_off"]) and np.array_equal(r["coline_pairs"], ln["coline_pairs"])
        assert (f1 - f0, h1 - h0) == (1, 0), "the frame was handed to the host tail"
    finally:
        ctx.close()


def test_batch_equals_single(pkg, synth):
    H, W = 96, 128
    imgs = [synth.make_frame(H, W, 40 + k, 0, "sparse" if k % 2 else "dense") for k in range(4)]
    one = get_ctx(pkg, W, H)
    singles = [one.extract(i) for i in imgs]
    ctx = get_ctx(pkg, W, H, batch=4)
    for r, s in zip(ctx.extract_batch(imgs), singles):
        for k in ("xy", "score", "un", "out", "desc", "edge_idx", "conn_idx", "coline_pairs"):
            assert np.array_equal(r[k], s[k], equal_nan=True), k
    # partial batch and idempotence
    again = ctx.extract_batch(imgs[:3])
    for r, s in zip(again, singles[:3]):
        assert np.array_equal(r["xy"], s["xy"]) and np.array_equal(r["desc"], s["desc"])


def test_bench_batch_equals_single_vga(pkg, synth):
    """The bench configuration (32 VGA streams per step, 1024 key points): every frame of the batch equals the
    same frame extracted alone, bit for bit (batching must not change any result)."""
    H, W, B = 480, 640, 32
    imgs = [synth.make_frame(H, W, 700 + k, k % 5, "dense" if k % 3 else "sparse") for k in range(B)]
    ctx = get_ctx(pkg, W, H, 1024, batch=B)
    rs = ctx.extract_batch(imgs)
    one = get_ctx(pkg, W, H, 1024)
    for k in (0, 7, 16, 31):
        s = one.extract(imgs[k])
        for key in ("xy", "score", "un", "out", "desc", "edge_idx", "edge_score", "conn_idx", "coline_pairs"):
            assert np.array_equal(rs[k][key], s[key], equal_nan=True), (k, key)


def test_repeated_bench_batches_are_idempotent(pkg, synth):
    """Forty extractions of two alternating 16-frame VGA batches through one context (the size at which the fused first
    convolution hands its tiles out through a counter that every launch must leave at zero): each result equals the
    first one of its batch."""
    H, W, B = 480, 640, 16
    sets = [[synth.make_frame(H, W, 900 + 16 * s + k, k % 3, "dense" if k % 2 else "sparse") for k in range(B)] for s in range(2)]
    ctx = get_ctx(pkg, W, H, 1024, batch=B)
    first = [ctx.extract_batch(imgs) for imgs in sets]
    for rep in range(38):
        rs = ctx.extract_batch(sets[rep % 2])
        for k in (0, 5, 15):
            for key in ("xy", "desc", "edge_idx"):
                assert np.array_equal(rs[k][key], first[rep % 2][k][key]), (rep, k, key)


def test_degenerate_frames(pkg):
    H, W = 64, 96
    ctx = get_ctx(pkg, W, H)
    r = ctx.extract(np.full((H, W), 128, np.uint8))  # no junctions: N = 0, empty graph
    assert r["n"] == 0 and r["n_edges"] == 0 and list(r["conn_off"]) == [0]
    # fewer than 10 key points: all-zero descriptors (PPGExtractor.cpp:520-524)
    img = np.full((H, W), 128, np.uint8)
    img[20:44, 30:66] = 255
    r = ctx.extract(img)
    assert 0 < r["n"] < 10 and not r["desc"].any()


def test_create_rejects_bad_arguments(pkg):
    with pytest.raises(pkg.PPGError, match="multiples of 8"):
        pkg.Context(pkg.default_config(100, 60))
    with pytest.raises(pkg.PPGError, match="cannot open"):
        pkg.Context(pkg.default_config(64, 64, weights="/nonexistent.bin"))
    ctx = get_ctx(pkg, 96, 64)
    with pytest.raises(ValueError):
        ctx.extract(np.zeros((10, 10), np.uint8))
    with pytest.raises(pkg.PPGError, match="max_batch"):
        ctx.extract_batch([np.zeros((64, 96), np.uint8)] * 2)


# ------------------------------- matching -------------------------------

def unit_rows(rng, n):
    a = rng.normal(size=(n, 256)).astype(np.float32)
    return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)


def test_dist_matrix_bit_exact(pkg, orc):
    rng = np.random.default_rng(1)
    ctx = get_ctx(pkg, 96, 64)
    for nA, nB in ((1, 1), (7, 130), (200, 65), (64, 64)):
        A, B = unit_rows(rng, nA), unit_rows(rng, nB)
        D = ctx.dist_matrix(A, B)
        for i, j in {(0, 0), (nA - 1, nB - 1), (nA // 2, nB // 3)}:
            assert D[i, j] == np.float32(orc.descriptor_distance(A[i], B[j]))
        ref = np.sqrt(((A[:, None, :].astype(np.float64) - B[None]) ** 2).sum(-1))
        assert float(np.abs(D - ref).max()) < 1e-6


def make_pair(rng, n, noise=0.05, shuffle=True):
    A = unit_rows(rng, n)
    perm = rng.permutation(n) if shuffle else np.arange(n)
    B = A[perm] + noise * rng.normal(size=(n, 256)).astype(np.float32) / 16
    B = (B / np.linalg.norm(B, axis=1, keepdims=True)).astype(np.float32)
    return A, B, perm


def test_match_plain_vs_oracle(pkg, orc):
    rng = np.random.default_rng(2)
    ctx = get_ctx(pkg, 96, 64)
    A, B, _ = make_pair(rng, 150)
    B[17] = B[3]  # exact tie: the first candidate in order must win
    skip = (rng.random(150) < 0.1).astype(np.uint8)
    off = [0]
    idx = []
    for i in range(150):
        k = int(rng.integers(0, 40))
        idx += list(rng.choice(150, size=k, replace=False))
        off.append(len(idx))
    m, best, second, _ = ctx.match_nn(A, B, np.array(off), np.array(idx), skip, mode=pkg.MATCH_PLAIN)
    import ctypes as C
    for i in range(150):
        cand = np.array(idx[off[i]:off[i + 1]], np.int32)
        b, s = C.c_float(), C.c_float()
        bi = orc.lib().ppgo_window_nn(A[i].ctypes.data_as(C.c_void_p), 256, B.ctypes.data_as(C.c_void_p),
                                      cand.ctypes.data_as(C.c_void_p), len(cand), skip.ctypes.data_as(C.c_void_p),
                                      C.byref(b), C.byref(s))
        assert (m[i], best[i], second[i]) == (bi, np.float32(b.value), np.float32(s.value)), i
    # dense (all pairs) variant
    m2, best2, _, _ = ctx.match_nn(A, B, mode=pkg.MATCH_PLAIN)
    D = ctx.dist_matrix(A, B)
    assert np.array_equal(m2, D.argmin(1)) and np.array_equal(best2, D.min(1))


@pytest.mark.parametrize("n,window", [(60, 0), (500, 0), (1024, 0), (300, 50), (1024, 50), (2048, 0), (3000, 80), (8200, 30)])
def test_search_for_initialization_vs_oracle(n, window, pkg, orc):
    rng = np.random.default_rng(3 + n)
    ctx = get_ctx(pkg, 640, 480)
    cfg = orc.default_cfg(640, 480)
    A, B, perm = make_pair(rng, n, noise=0.3)
    # duplicates and near-duplicates so that claims and the ratio test both fire
    B[5] = B[6]
    A[10] = A[11]
    pos1 = np.stack([rng.uniform(5, 634, n), rng.uniform(5, 474, n)], 1).astype(np.float32)
    pos2 = np.empty_like(pos1)
    pos2[:] = pos1[perm] + rng.uniform(-20, 20, (n, 2)).astype(np.float32)
    cnt_o, m_o, prev_o = orc.search_for_initialization(cfg, 0.9, A, B, pos2, pos1, window)
    cnt_g, m_g, prev_g = ctx.search_for_initialization(0.9, A, B, pos2, pos1, window)
    assert cnt_g == cnt_o and np.array_equal(m_g, m_o) and np.array_equal(prev_g, prev_o)
    assert cnt_o > n // 4


def test_match_claims_force_fallback(pkg, orc):
    """Many F1 rows share the same few nearest F2 rows, so the 8-entry lists run dry and the resolver has
    to rescan rows; the result must still be the sequential one."""
    rng = np.random.default_rng(9)
    ctx = get_ctx(pkg, 640, 480)
    cfg = orc.default_cfg(640, 480)
    base = unit_rows(rng, 4)
    A = np.repeat(base, 40, 0) + 0.02 * rng.normal(size=(160, 256)).astype(np.float32)
    A = (A / np.linalg.norm(A, axis=1, keepdims=True)).astype(np.float32)
    B = np.repeat(base, 30, 0) + 0.02 * rng.normal(size=(120, 256)).astype(np.float32)
    B = (B / np.linalg.norm(B, axis=1, keepdims=True)).astype(np.float32)
    pos = np.zeros((160, 2), np.float32)
    pos2 = np.zeros((120, 2), np.float32)
    cnt_o, m_o, _ = orc.search_for_initialization(cfg, 1.1, A, B, pos2, pos, 0)
    cnt_g, m_g, _ = ctx.search_for_initialization(1.1, A, B, pos2, pos, 0)
    assert cnt_g == cnt_o and np.array_equal(m_g, m_o)
    assert cnt_o > 60


def test_windowed_claims_contention(pkg, orc):
    """The windowed claim pass (candidate lists, fixed-point iteration on the device) under heavy contention: clustered
    descriptors whose rows all lie inside each other's 50-pixel windows, so that most rows find their nearest columns
    claimed by earlier rows and the iteration needs many rounds; nnratio above and below 1."""
    rng = np.random.default_rng(19)
    ctx = get_ctx(pkg, 640, 480)
    cfg = orc.default_cfg(640, 480)
    base = unit_rows(rng, 6)
    A = np.repeat(base, 50, 0) + 0.02 * rng.normal(size=(300, 256)).astype(np.float32)
    A = (A / np.linalg.norm(A, axis=1, keepdims=True)).astype(np.float32)
    B = np.repeat(base, 40, 0) + 0.02 * rng.normal(size=(240, 256)).astype(np.float32)
    B = (B / np.linalg.norm(B, axis=1, keepdims=True)).astype(np.float32)
    B[7] = B[8]
    pos = (np.array([300.0, 200.0]) + rng.uniform(-30, 30, (300, 2))).astype(np.float32)
    pos2 = (np.array([300.0, 200.0]) + rng.uniform(-30, 30, (240, 2))).astype(np.float32)
    counts = []
    for ratio in (1.1, 0.95):
        cnt_o, m_o, prev_o = orc.search_for_initialization(cfg, ratio, A, B, pos2, pos, 50)
        cnt_g, m_g, prev_g = ctx.search_for_initialization(ratio, A, B, pos2, pos, 50)
        assert cnt_g == cnt_o and np.array_equal(m_g, m_o) and np.array_equal(prev_g, prev_o), ratio
        counts.append(cnt_o)
    assert counts[0] > 100 and counts[1] > 5


def test_match_frames_resident(pkg, orc, synth):
    H, W = 480, 640
    ctx = get_ctx(pkg, W, H, 1024, batch=2)
    imgs = [synth.make_frame(H, W, 11, t, "dense") for t in (0, 1)]
    r = ctx.extract_batch(imgs)
    cnt, m12 = ctx.match_frames(0, 1, r[0]["n"], 0.9)
    cfg = orc.default_cfg(W, H, 1024)
    cnt_o, m_o, _ = orc.search_for_initialization(cfg, 0.9, r[0]["desc"], r[1]["desc"], r[1]["un"], r[0]["un"], 0)
    assert cnt == cnt_o and np.array_equal(m12, m_o)
    assert cnt > 400


def test_match_prev_batch_resident(pkg, orc, synth):
    """Two consecutive extract calls of 3 streams; every stream's frame t is matched against t+1 on the GPU."""
    H, W = 96, 128
    ctx = get_ctx(pkg, W, H, 500, batch=4)
    cfg = orc.default_cfg(W, H)
    scenes = [(1, "sparse"), (7, "dense"), (3, "sparse")]
    r0 = ctx.extract_batch([synth.make_frame(H, W, s, 0, k) for s, k in scenes])
    r1 = ctx.extract_batch([synth.make_frame(H, W, s, 1, k) for s, k in scenes])
    cnts, m12 = ctx.match_prev_batch(3, 0.9)
    for b in range(3):
        co, mo, _ = orc.search_for_initialization(cfg, 0.9, r0[b]["desc"], r1[b]["desc"], r1[b]["un"], r0[b]["un"], 0)
        assert cnts[b] == co and np.array_equal(m12[b, :r0[b]["n"]], mo) and np.all(m12[b, r0[b]["n"]:] == -1)
    assert cnts.sum() > 20
    with pytest.raises(pkg.PPGError):
        ctx.match_prev_batch(4, 0.9)


def test_match_prev_batch_resident_beyond_1024_keypoints(pkg, orc, synth):
    """JUNCTION_MAX_NUM = 2048: the resident frame-to-frame match leaves the matrix-free path (at most 1024 rows) for the
    batched distance matrix, the eight-nearest lists and the fixed-point claim pass; two streams of 768 x 1024 frames."""
    H, W = 768, 1024
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=2, max_num=2048))
    try:
        cfg = orc.default_cfg(W, H, 2048)
        scenes = [(11, "dense"), (12, "dense")]
        r0 = ctx.extract_batch([synth.make_frame(H, W, s, 0, k) for s, k in scenes])
        r1 = ctx.extract_batch([synth.make_frame(H, W, s, 1, k) for s, k in scenes])
        assert max(r["n"] for r in r0) > 1024, [r["n"] for r in r0]
        cnts, m12 = ctx.match_prev_batch(2, 0.9)
        for b in range(2):
            co, mo, _ = orc.search_for_initialization(cfg, 0.9, r0[b]["desc"], r1[b]["desc"], r1[b]["un"], r0[b]["un"], 0)
            assert cnts[b] == co and np.array_equal(m12[b, :r0[b]["n"]], mo) and np.all(m12[b, r0[b]["n"]:] == -1)
        assert cnts.sum() > 800
    finally:
        ctx.close()


def test_pipelined_submit_wait_equals_blocking(pkg, synth):
    """Four batches in flight: results (frames and matches) must equal the blocking calls bit for bit."""
    H, W = 96, 128
    ctx = get_ctx(pkg, W, H, 500, batch=4)
    seqs = [[synth.make_frame(H, W, s, t, k) for s, k in ((1, "sparse"), (7, "dense"), (3, "sparse"))] for t in range(4)]
    ref_frames, ref_matches = [], []
    for t in range(4):
        ref_frames.append(ctx.extract_batch(seqs[t]))
        if t:
            ref_matches.append(ctx.match_prev_batch(3, 0.9))
    got = []
    ctx.submit_batch(seqs[0], match_prev=False)
    for t in range(1, 4):
        ctx.submit_batch(seqs[t], match_prev=True)
    # four batches in flight: the ring is full
    with pytest.raises(pkg.PPGError, match="in flight"):
        ctx.submit_batch(seqs[0])
    with pytest.raises(pkg.PPGError, match="pending"):
        ctx.extract(seqs[0][0])
    for t in range(4):
        got.append(ctx.wait_batch())
    with pytest.raises(pkg.PPGError, match="nothing was submitted"):
        ctx.wait_batch()
    for t in range(4):
        frames, cnts, m12, n = got[t]
        assert n == 3
        for r, s in zip(frames, ref_frames[t]):
            for k in ("xy", "score", "desc", "edge_idx", "conn_idx", "coline_pairs"):
                assert np.array_equal(r[k], s[k], equal_nan=True), (t, k)
        if t:
            assert np.array_equal(cnts, ref_matches[t - 1][0]) and np.array_equal(m12, ref_matches[t - 1][1])
        else:
            assert not cnts.any() and np.all(m12 == -1)


def test_grid_matches_oracle(pkg, orc):
    rng = np.random.default_rng(4)
    ctx = get_ctx(pkg, 640, 480)
    cfg = orc.default_cfg(640, 480)
    pos = np.stack([rng.uniform(0, 640, 400), rng.uniform(0, 480, 400)], 1).astype(np.float32)
    for _ in range(50):
        x, y, r = float(rng.uniform(-20, 660)), float(rng.uniform(-20, 500)), float(rng.uniform(1, 80))
        assert np.array_equal(ctx.features_in_area(pos, x, y, r), orc.features_in_area(cfg, pos, x, y, r))


def random_map_graph(rng, desc, pos, m, n_medges, key_edges=None):
    import importlib
    return importlib.import_module("ppg_slam_amd.synth").random_map_graph(rng, desc, pos, m, n_medges, key_edges)


def test_extend_map_matches_vs_oracle(pkg, orc, synth):
    H, W = 480, 640
    ctx = get_ctx(pkg, W, H)
    cfg = orc.default_cfg(W, H)
    r = ctx.extract(synth.make_frame(H, W, 1, 0, "sparse"))
    rng = np.random.default_rng(5)
    for trial in range(3):
        m = 300
        g = random_map_graph(rng, r["desc"], r["un"], m, 400)
        tracked = (rng.random(m) < 0.05).astype(np.uint8)
        fmp = np.full(r["n"], -1, np.int32)
        pre = rng.choice(r["n"], size=30, replace=False)
        fmp[pre] = rng.integers(0, m, 30)
        fme = np.full(r["n_edges"], -1, np.int32)
        args = (0.8, 3.0 + 2 * trial, g, r["desc"], r["un"], r["edge_idx"], r["conn_off"], r["conn_idx"], tracked, fmp, fme)
        co, to, po, eo = orc.extend_map_matches(cfg, *args)
        cg, tg, pg, eg = ctx.extend_map_matches(*args)
        assert cg == co and np.array_equal(tg, to) and np.array_equal(pg, po) and np.array_equal(eg, eo)
        assert co > 50 and (eo >= 0).sum() > 0


def test_extend_map_matches_fuzz(pkg, orc, synth):
    """Many small random problems for the point-pair-graph matcher, with the degenerate shapes a real map can hold: map edges
    from a point to itself, repeated edges, missing other ends, points without edges, everything pre-tracked or nothing,
    slots pre-assigned to bad or unobserved points, windows of every size including none, nnratio above and below 1."""
    H, W = 240, 320
    ctx = get_ctx(pkg, W, H, 500)
    cfg = orc.default_cfg(W, H, 500)
    r = ctx.extract(synth.make_frame(H, W, 77, 0, "dense"))
    n = r["n"]
    assert n > 150 and r["n_edges"] > 50
    rng = np.random.default_rng(4242)
    hits = 0
    for trial in range(40):
        m = int(rng.integers(1, 180))
        g = random_map_graph(rng, r["desc"], r["un"], min(m, n), int(rng.integers(0, 4 * m + 1)),
                             r["edge_idx"] if trial % 3 == 0 else None)
        m = g["desc"].shape[0]
        ne = len(g["edge_other"])
        if ne:
            # self-loops and repeated other ends
            owner = np.repeat(np.arange(m), np.diff(g["edge_off"]))
            loops = rng.random(ne) < 0.05
            g["edge_other"][loops] = owner[loops]
            dup = rng.random(ne) < 0.1
            g["edge_other"][dup] = int(rng.integers(0, m))
        g["in_view"] = (rng.random(m) < rng.uniform(0.3, 1.0)).astype(np.uint8)
        g["bad"] = (rng.random(m) < rng.uniform(0.0, 0.3)).astype(np.uint8)
        g["nobs"] = rng.integers(0, 2, m).astype(np.int32) * rng.integers(0, 4, m).astype(np.int32)
        mode = trial % 4
        tracked = (np.ones(m) if mode == 1 else rng.random(m) < (0.0 if mode == 0 else 0.3)).astype(np.uint8)
        fmp = np.full(n, -1, np.int32)
        k = int(rng.integers(0, n // 2))
        if k:
            fmp[rng.choice(n, size=k, replace=False)] = rng.integers(0, m, k)
        fme = np.full(r["n_edges"], -1, np.int32)
        th = float(rng.choice([0.5, 2.0, 5.0, 15.0, 60.0]))
        ratio = float(rng.choice([0.6, 0.9, 1.2]))
        args = (ratio, th, g, r["desc"], r["un"], r["edge_idx"], r["conn_off"], r["conn_idx"], tracked, fmp, fme)
        co, to, po, eo = orc.extend_map_matches(cfg, *args)
        cg, tg, pg, eg = ctx.extend_map_matches(*args)
        assert cg == co and np.array_equal(tg, to) and np.array_equal(pg, po) and np.array_equal(eg, eo), (trial, m, ne, th, ratio)
        hits += co
    assert hits > 500


def test_match_projected_fuzz(pkg, orc):
    """The windowed search shared by SearchByProjection and Fuse (candidate lists from the grid, first argmin, threshold
    and ratio, accepted rows claiming their slot: a fixed-point iteration on the device) against the sequential pass
    written out here row by row, the distances of every window taken from the oracle's chain: random windows (none to
    everything), ties, blocked slots, rows that do not claim, rows switched off, the reprojection gate."""
    import ctypes as C
    W, H = 640, 480
    ctx = get_ctx(pkg, W, H)
    rng = np.random.default_rng(99)
    total = 0
    for trial in range(25):
        n1, n2 = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        B = unit_rows(rng, n2)
        src = rng.integers(0, n2, n1)
        A = B[src] + float(rng.choice([0.05, 0.3, 1.0])) * rng.normal(size=(n1, 256)).astype(np.float32) / 16
        A = (A / np.linalg.norm(A, axis=1, keepdims=True)).astype(np.float32)
        if n2 > 4:
            B[3] = B[1]                      # exact ties: the first candidate in window order wins
        pos2 = np.stack([rng.uniform(0, W - 1, n2), rng.uniform(0, H - 1, n2)], 1).astype(np.float32)
        uv = (pos2[src] + rng.uniform(-12, 12, (n1, 2))).astype(np.float32)
        radius = rng.choice([0.0, 3.0, 10.0, 40.0, 1000.0], n1).astype(np.float32)
        row_ok = (rng.random(n1) < 0.9).astype(np.uint8) if trial % 2 else None
        claims = (rng.random(n1) < 0.7).astype(np.uint8) if trial % 3 else None
        blocked = (rng.random(n2) < 0.2).astype(np.uint8) if (trial % 3 or trial % 5 == 0) else None
        th = float(rng.choice([0.3, 0.7, 0.8, 2.0]))
        ratio = float(rng.choice([0.0, 0.6, 0.9]))
        e2 = float(rng.choice([-1.0, 36.0]))
        cnt, bi, bd, blk = ctx.match_projected(pos2, B, A, uv, radius, th, ratio, row_ok, claims, blocked, e2)
        # the sequential pass
        state = blocked.copy() if blocked is not None else np.zeros(n2, np.uint8)
        want_i, want_d, nm = np.full(n1, -1, np.int32), np.full(n1, 1e6, np.float32), 0
        for i in range(n1):
            if row_ok is not None and not row_ok[i]:
                continue
            cand = ctx.features_in_area(pos2, float(uv[i, 0]), float(uv[i, 1]), float(radius[i]))
            if e2 >= 0:
                d2 = (np.float32(uv[i, 0]) - pos2[cand, 0]) ** 2 + (np.float32(uv[i, 1]) - pos2[cand, 1]) ** 2
                cand = cand[~(d2 > np.float32(e2))]
            cand = np.ascontiguousarray(cand, np.int32)
            b, s2 = C.c_float(), C.c_float()
            j = orc.lib().ppgo_window_nn(A[i].ctypes.data_as(C.c_void_p), 256, B.ctypes.data_as(C.c_void_p),
                                         cand.ctypes.data_as(C.c_void_p), len(cand), state.ctypes.data_as(C.c_void_p),
                                         C.byref(b), C.byref(s2))
            want_d[i] = b.value
            if b.value <= th and not (ratio > 0 and b.value > np.float32(ratio) * np.float32(s2.value)):
                want_i[i] = j
                nm += 1
                if claims is not None and blocked is not None and claims[i]:
                    state[j] = 1
        tag = (trial, n1, n2, th, ratio, e2)
        assert np.array_equal(bi, want_i), tag
        assert np.array_equal(bd, want_d), tag
        assert cnt == nm, tag
        if blocked is not None and claims is not None:
            assert np.array_equal(blk, state), tag
        total += nm
    assert total > 500


def test_match_nn_claim_mode_fuzz(pkg, orc):
    """ppg_match_nn in its claim mode (the loop of SearchForInitialization) over random candidate lists: empty rows, rows
    with fewer and with many more than eight candidates, a candidate listed twice, skipped columns, clustered descriptors
    (most rows find their nearest columns taken), against the sequential pass written out here."""
    import ctypes as C
    ctx = get_ctx(pkg, 640, 480)
    rng = np.random.default_rng(123)
    total = 0
    for trial in range(25):
        nA, nB = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        k = int(rng.integers(1, 12))
        base = unit_rows(rng, k)
        mk = lambda n: base[rng.integers(0, k, n)] + float(rng.choice([0.02, 0.2])) * rng.normal(size=(n, 256)).astype(np.float32) / 16
        A, B = mk(nA), mk(nB)
        A = (A / np.linalg.norm(A, axis=1, keepdims=True)).astype(np.float32)
        B = (B / np.linalg.norm(B, axis=1, keepdims=True)).astype(np.float32)
        off, idx = [0], []
        for i in range(nA):
            c = int(rng.choice([0, 1, 3, 8, 9, 40, nB]))
            c = min(c, nB)
            row = list(rng.choice(nB, size=c, replace=False)) if c else []
            if len(row) > 2 and rng.random() < 0.3:
                row.append(row[0])              # the same column listed twice
            idx += row
            off.append(len(idx))
        off = np.array(off, np.int32)
        idx = np.array(idx if idx else [0], np.int32)
        skip = (rng.random(nB) < 0.15).astype(np.uint8) if trial % 2 else None
        th, ratio = float(rng.choice([0.3, 0.7, 2.0])), float(rng.choice([0.8, 0.95, 1.1]))
        m, best, second, cnt = ctx.match_nn(A, B, off, idx, skip, th=th, ratio=ratio, mode=pkg.MATCH_INIT)
        state = skip.copy() if skip is not None else np.zeros(nB, np.uint8)
        want = np.full(nA, -1, np.int32)
        nm = 0
        for i in range(nA):
            cand = np.ascontiguousarray(idx[off[i]:off[i + 1]], np.int32)
            b, s2 = C.c_float(), C.c_float()
            j = orc.lib().ppgo_window_nn(A[i].ctypes.data_as(C.c_void_p), 256, B.ctypes.data_as(C.c_void_p),
                                         cand.ctypes.data_as(C.c_void_p), len(cand), state.ctypes.data_as(C.c_void_p),
                                         C.byref(b), C.byref(s2))
            assert (best[i], second[i]) == (np.float32(b.value), np.float32(s2.value)), (trial, i)
            if j >= 0 and b.value <= th and np.float32(b.value) < np.float32(s2.value) * np.float32(ratio):   # Matcher.cpp:630
                want[i] = j
                state[j] = 1
                nm += 1
        assert np.array_equal(m, want) and cnt == nm, (trial, nA, nB, th, ratio)
        total += nm
    assert total > 300


def test_nms_fuzz_adversarial_score_maps(pkg, orc):
    """Greedy radius-4 NMS as a fixed point on the device against the oracle's sorted sequential pass, on score maps made
    to be hard: plateaus of equal scores (the order of ties is y*W + x), values quantised to a few levels, checkerboards,
    ridges, everything above the threshold, peaks on the four-pixel border, and caps from 1 to more than the survivors."""
    rng = np.random.default_rng(31)
    n_total = 0
    for trial in range(24):
        H, W = [(96, 128), (120, 160), (64, 200), (240, 320)][trial % 4]
        kind = trial % 6
        if kind == 0:
            sc = np.full((H, W), 0.5, np.float32)                                  # one plateau
        elif kind == 1:
            sc = (rng.integers(0, 4, (H, W)) / 4.0).astype(np.float32)             # four levels: ties everywhere
        elif kind == 2:
            yy, xx = np.mgrid[0:H, 0:W]
            sc = (((yy // 3 + xx // 3) % 2) * 0.7 + 0.01 * rng.random((H, W))).astype(np.float32)
        elif kind == 3:
            sc = np.zeros((H, W), np.float32)                                      # ridges of equal height
            sc[::7, :] = 0.6
            sc[:, ::11] = 0.6
        elif kind == 4:
            sc = rng.random((H, W)).astype(np.float32)                             # everything is a candidate
        else:
            sc = np.zeros((H, W), np.float32)                                      # sparse peaks, some inside the border band
            for _ in range(300):
                sc[int(rng.integers(0, H)), int(rng.integers(0, W))] = float(rng.choice([0.3, 0.3, 0.9, rng.random()]))
        cap = int(rng.choice([1, 7, 60, 500, 1024]))
        ctx = pkg.Context(pkg.default_config(W, H, max_num=cap))
        try:
            ctx.inference([np.zeros((H, W), np.uint8)])
            ctx.write_tensor(pkg.T_SCORE, sc)
            ctx.detect_keypoints(1)
            ctx.detect_lines(1)
            ctx.gen_descriptors(1)
            r = ctx.collect(1)[0]
        finally:
            ctx.close()
        o = orc.detect_keypoints(orc.default_cfg(W, H, cap), sc)
        tag = (trial, H, W, kind, cap)
        assert r["n"] == o["n"], tag
        assert np.array_equal(r["xy"], o["xy"]) and np.array_equal(r["score"], o["score"]), tag
        assert np.array_equal(r["un"], o["un"]) and np.array_equal(r["out"], o["out"]), tag
        n_total += r["n"]
    assert n_total > 2000


def test_line_graph_fuzz_drawn_wireframes(pkg, orc):
    """refineHeatMap + detectLines on the device against the oracle on injected maps: random wireframes (key points as
    score peaks, the heat map drawn along random segments between them, with noise and gaps), which make for many
    candidates that overlap at shared end points, collinear continuations through a point, short edges and stars of many
    lines at one point.  Every integer output must be equal, the edge scores bit for bit."""
    rng = np.random.default_rng(57)
    edges_total = coline_total = 0
    for trial in range(12):
        H, W = [(240, 320), (200, 264), (480, 640)][trial % 3]
        npts = int(rng.choice([12, 40, 120, 300]))
        pts = np.stack([rng.integers(6, W - 6, npts), rng.integers(6, H - 6, npts)], 1)
        if trial % 2:                                              # points on common lines: collinear pairs
            for k in range(0, npts - 2, 3):
                t = rng.uniform(0.3, 0.7)
                pts[k + 2] = np.round(pts[k] + t * (pts[k + 1] - pts[k])).astype(int)
        score = np.zeros((H, W), np.float32)
        for k, (x, y) in enumerate(pts):
            score[y, x] = 0.5 + 0.4 * rng.random()
        heat = (0.05 * rng.random((H, W))).astype(np.float32)
        nseg = int(npts * rng.uniform(0.8, 2.5))
        for _ in range(nseg):
            a, b = rng.choice(npts, 2, replace=False)
            if trial % 4 == 0 and rng.random() < 0.5:
                a = 0                                              # a star around the first point
            (x0, y0), (x1, y1) = pts[a], pts[b]
            L = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
            xs = np.round(np.linspace(x0, x1, L)).astype(int)
            ys = np.round(np.linspace(y0, y1, L)).astype(int)
            v = (0.6 + 0.4 * rng.random(L)).astype(np.float32)
            if rng.random() < 0.3:
                g0 = int(rng.integers(0, L))
                v[g0:g0 + int(rng.integers(1, 6))] = 0.0           # a gap
            heat[ys, xs] = np.maximum(heat[ys, xs], v)
            heat[np.clip(ys + 1, 0, H - 1), xs] = np.maximum(heat[np.clip(ys + 1, 0, H - 1), xs], 0.5 * v)
        cap = 500
        ctx = pkg.Context(pkg.default_config(W, H, max_num=cap))
        try:
            ctx.inference([np.zeros((H, W), np.uint8)])
            ctx.write_tensor(pkg.T_SCORE, score)
            ctx.write_tensor(pkg.T_HEAT, heat)
            ctx.detect_keypoints(1)
            ctx.detect_lines(1)
            ctx.gen_descriptors(1)
            r = ctx.collect(1)[0]
            refined = ctx.read_tensor(pkg.T_HEAT_REFINED)
        finally:
            ctx.close()
        cfg = orc.default_cfg(W, H, cap)
        kp = orc.detect_keypoints(cfg, score)
        assert np.array_equal(r["xy"], kp["xy"])
        ref_o = orc.refine_heat(cfg, heat)
        assert np.array_equal(refined, ref_o), trial
        ln = orc.detect_lines(cfg, ref_o, kp["un"], kp["out"])
        tag = (trial, H, W, npts, nseg)
        assert np.array_equal(r["edge_idx"], ln["edge_idx"]), tag
        assert np.array_equal(r["edge_score"], ln["edge_score"], equal_nan=True), tag
        assert np.array_equal(r["conn_off"], ln["conn_off"][:r["n"] + 1]) and np.array_equal(r["conn_idx"], ln["conn_idx"][:len(r["conn_idx"])]), tag
        assert np.array_equal(r["coline_pairs"], ln["coline_pairs"][:len(r["coline_pairs"])]), tag
        edges_total += r["n_edges"]
        coline_total += len(r["coline_pairs"])
    print(f"wireframes: {edges_total} edges, {coline_total} collinear pairs")
    assert edges_total > 300 and coline_total > 10


def test_host_mirror_program(pkg, synth, tmp_path):
    """The C++ mirror (PPGExtractor::run + Matcher::SearchForInitialization) gives what the C ABI gives."""
    import os
    import subprocess
    from conftest import ROOT
    demo = os.path.join(ROOT, "ppg-slam_amd", "host", "shim_demo")
    assert os.path.exists(demo), "build it with make -C ppg-slam_amd/csrc demo"
    H, W = 96, 128
    imgs = [synth.make_frame(H, W, 1, t, "sparse") for t in (0, 1)]
    paths = []
    for k, im in enumerate(imgs):
        p = tmp_path / f"img{k}.raw"
        p.write_bytes(im.tobytes())
        paths.append(str(p))
    out = subprocess.run([demo, os.path.dirname(pkg.DEFAULT_WEIGHTS), paths[0], paths[1], str(W), str(H)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    # (the result line of the demo; RCCL prints a banner of its own to stdout when the demo's barrier section sets it up)
    line = [ln for ln in out.stdout.splitlines() if re.fullmatch(r"(-?\d+ ){5}-?\d+", ln.strip())][-1]
    n0, e0, n1, conn0, nm, chk = [int(v) for v in line.split()]
    ctx = get_ctx(pkg, W, H)
    r0, r1 = ctx.extract(imgs[0]), ctx.extract(imgs[1])
    cnt, m12, _ = ctx.search_for_initialization(0.9, r0["desc"], r1["desc"], r1["un"], r0["un"], 50)
    ref = 0
    for v in m12:
        ref = (ref * 31 + (int(v) + 2)) & 0xFFFFFFFFFFFFFFFF
    if ref >= 1 << 63:
        ref -= 1 << 64
    assert (n0, e0, n1, conn0, nm) == (r0["n"], r0["n_edges"], r1["n"], len(r0["conn_idx"]), cnt)
    assert chk == ref


def _read_dump(path):
    import struct
    data = open(path, "rb").read()
    ofs = 0
    dtypes = [np.int32, np.float32, np.float32,                                       # counts, K, pose
              np.float32, np.float32, np.float32, np.float32, np.float32, np.uint8, np.int32,   # map
              np.int32, np.uint8, np.float32, np.float32,                               # last_mp, outlier, pos B, desc B
              np.int32, np.int32, np.int32, np.int32, np.int32, np.int32, np.uint8, np.int32, np.int32,
              np.int32, np.int32, np.int32, np.int32, np.int32, np.int32, np.int32,    # counts2, FeatureVectors A and B
              np.float32, np.int32, np.int32, np.int32, np.int32,                      # pos A, mp A, mp B, bow results
              np.int32, np.int32, np.int32, np.int32]                                  # triangulation inputs / pairs, root
    out = []
    for dt in dtypes:
        (n,) = struct.unpack_from("<i", data, ofs)
        ofs += 4
        nbytes = n * np.dtype(dt).itemsize
        out.append(np.frombuffer(data, dt, n, ofs).copy())
        ofs += nbytes
    assert ofs == len(data)
    return out


def test_host_mirror_projection_matchers_vs_oracle(pkg, orc, synth, tmp_path):
    """Matcher::SearchByProjection (frame/frame and key-frame/Sim3) and both Fuse variants of the C++ mirror, which
    take every candidate distance from the GPU, against the line-by-line CPU restatement on the same synthetic map."""
    import os
    import subprocess
    from conftest import ROOT
    demo = os.path.join(ROOT, "ppg-slam_amd", "host", "shim_demo")
    assert os.path.exists(demo), "build it with make -C ppg-slam_amd/csrc demo"
    H, W = 480, 640
    imgs = [synth.make_frame(H, W, 61, t, "sparse") for t in (0, 1)]
    paths = []
    for k, im in enumerate(imgs):
        p = tmp_path / f"img{k}.raw"
        p.write_bytes(im.tobytes())
        paths.append(str(p))
    dump = str(tmp_path / "proj.bin")
    out = subprocess.run([demo, os.path.dirname(pkg.DEFAULT_WEIGHTS), paths[0], paths[1], str(W), str(H), dump],
                         capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr
    (counts, K, pose, pos, normal, dmin, dmax, desc, bad, nobs, last_mp, outlier, posb, descb, cur_before, cur_after,
     points, matched_before, matched_after, fuse_pts, in_kf, best_a, best_b,
     counts2, nodeA, offA, idxA, nodeB, offB, idxB, posa, mpA, mpB, bow_f, bow_kk, mpA2, mpB2, tri, bow_root) = _read_dump(dump)
    n1, n2, n3, n4, Wd, Hd = [int(v) for v in counts]
    assert (Wd, Hd) == (W, H) and len(bad) > 100 and len(posb) // 2 > 100
    cfg = orc.default_cfg(W, H)
    for k in range(9):
        cfg.K[k] = float(K[k])
    mp = {"pos": pos.reshape(-1, 3), "normal": normal.reshape(-1, 3), "min_dist": dmin, "max_dist": dmax,
          "desc": desc.reshape(-1, 256), "bad": bad, "nobs": nobs}
    R, t = pose[:9], pose[9:]
    posb = posb.reshape(-1, 2)
    descb = descb.reshape(-1, 256)
    o1, cur = orc.search_by_projection_frames(cfg, R, t, mp, last_mp, outlier, posb, descb, cur_before, 15.0)
    assert o1 == n1 and o1 > 20 and np.array_equal(cur, cur_after)
    o2, m = orc.search_by_projection_sim3(cfg, R, t, mp, points, posb, descb, matched_before, 10, 1.0)
    assert o2 == n2 and o2 > 20 and np.array_equal(m, matched_after)
    o3, ba = orc.fuse_match(cfg, R, t, mp, fuse_pts, in_kf, posb, descb, 3.0, True)
    assert o3 == n3 and o3 > 5 and np.array_equal(ba, best_a)
    o4, bb = orc.fuse_match(cfg, R, t, mp, fuse_pts, in_kf, posb, descb, 4.0, False)
    assert o4 == n4 and o4 >= o3 and np.array_equal(bb, best_b)
    # ---- the FeatureVector matchers (desc of frame A = the map descriptors, by construction of the demo map) ----
    n5, n6, n7, n8 = [int(v) for v in counts2]
    fvA = [(int(nodeA[k]), idxA[offA[k]:offA[k + 1]].tolist()) for k in range(len(nodeA))]
    fvB = [(int(nodeB[k]), idxB[offB[k]:offB[k + 1]].tolist()) for k in range(len(nodeB))]
    assert len(fvA) >= 3 and len(fvB) >= 3 and [n for n, _ in fvA] != [n for n, _ in fvB]
    desca = desc.reshape(-1, 256)
    o5, out = orc.search_by_bow_frame(cfg, 0.9, fvA, fvB, desca, mpA, bad, descb)
    assert o5 == n5 and o5 > 10 and np.array_equal(out, bow_f)
    o6, out = orc.search_by_bow_keyframes(cfg, 0.9, fvA, fvB, desca, mpA, descb, mpB, bad)
    assert o6 == n6 and o6 > 5 and np.array_equal(out, bow_kk)
    o7, pairs = orc.search_for_triangulation(cfg, np.eye(3), np.zeros(3), R, t, fvA, fvB, desca, posa.reshape(-1, 2), mpA2,
                                             descb, posb, mpB2)
    assert o7 == n7 and o7 > 5 and np.array_equal(pairs.reshape(-1), tri)
    root = lambda n: [(0, list(range(n)))]
    o8, out = orc.search_by_bow_frame(cfg, 0.9, root(len(desca)), root(len(descb)), desca, mpA, bad, descb)
    assert o8 == n8 and o8 >= o5 and np.array_equal(out, bow_root)


def test_distinctive_descriptors_vs_oracle(pkg, orc):
    """MapPoint::ComputeDistinctiveDescriptors batched over 60 map points with 0..12 observations each."""
    rng = np.random.default_rng(5)
    ctx = get_ctx(pkg, 128, 96)
    segs, rows = [0], []
    for p in range(60):
        k = int(rng.integers(0, 13))
        centre = rng.standard_normal(256).astype(np.float32)
        for _ in range(k):
            v = centre + (0.15 if p % 3 else 2.0) * rng.standard_normal(256).astype(np.float32)
            rows.append(v / np.linalg.norm(v))
        segs.append(len(rows))
    desc = np.stack(rows).astype(np.float32)
    best = ctx.distinctive_descriptors(desc, segs)
    ref = [orc.distinctive_descriptor(desc[segs[p]:segs[p + 1]]) if segs[p + 1] > segs[p] else 0 for p in range(60)]
    assert best.tolist() == ref
    assert len(set(ref)) > 3


@pytest.mark.parametrize("fisheye,D", [(0, (-0.28, 0.07, 2e-4, 2e-5)), (1, (-0.01, 0.004, -0.002, 0.0005))])
def test_lens_distortion_path_vs_oracle(fisheye, D, pkg, orc, synth):
    """D[0] != 0 (EuRoC-like pinhole coefficients, a Kannala-Brandt set): key points are undistorted, the refined heat
    map goes through cv::remap before the line graph (PPGExtractor.cpp:221-223, 261-262).  Integer outputs exact
    against the oracle when both start from the GPU's CNN tensors."""
    H, W = 480, 640
    cfg = pkg.default_config(W, H, max_num=500)
    ocfg = orc.default_cfg(W, H, 500)
    cfg.fisheye = ocfg.fisheye = fisheye
    for k in range(4):
        cfg.D[k] = ocfg.D[k] = D[k]
    for k in range(9):
        ocfg.K[k] = cfg.K[k]
    ctx = pkg.Context(cfg)
    try:
        img = synth.make_frame(H, W, 91, 0, "sparse")
        r = ctx.extract(img)
        kp = orc.detect_keypoints(ocfg, ctx.read_tensor(pkg.T_SCORE))
        assert r["n"] == kp["n"] > 50
        assert np.array_equal(r["xy"], kp["xy"]) and np.array_equal(r["un"], kp["un"]) and np.array_equal(r["out"], kp["out"])
        assert float(np.abs(r["un"] - r["xy"]).max()) > 1.0          # the lens model does move points
        refined = orc.refine_heat(ocfg, ctx.read_tensor(pkg.T_HEAT))
        assert np.array_equal(ctx.read_tensor(pkg.T_HEAT_REFINED), refined)
        remapped = orc.remap_heat(ocfg, refined)
        assert np.array_equal(ctx.read_tensor(pkg.T_HEAT_REMAPPED), remapped)
        assert not np.array_equal(remapped, refined)
        ln = orc.detect_lines(ocfg, remapped, kp["un"], kp["out"])
        assert np.array_equal(r["edge_idx"], ln["edge_idx"]) and len(ln["edge_idx"]) > 10
        assert np.array_equal(r["edge_score"], ln["edge_score"], equal_nan=True)
        assert np.array_equal(r["conn_idx"], ln["conn_idx"]) and np.array_equal(r["coline_pairs"], ln["coline_pairs"])
    finally:
        ctx.close()


def test_match_mutual_vs_oracle(pkg, orc):
    """Mutual nearest neighbours (the agreement check of Matcher::SearchBySim3): all pairs and with candidate windows."""
    rng = np.random.default_rng(9)
    ctx = get_ctx(pkg, 128, 96)
    nA, nB = 300, 260
    base = rng.standard_normal((320, 256)).astype(np.float32)
    A = base[:nA] + 0.25 * rng.standard_normal((nA, 256)).astype(np.float32)
    B = base[rng.permutation(320)[:nB]] + 0.25 * rng.standard_normal((nB, 256)).astype(np.float32)
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    B /= np.linalg.norm(B, axis=1, keepdims=True)
    n, m12 = ctx.match_mutual(A, B, 0.8)
    on, om = orc.mutual_nn(A, B, 0.8)
    assert n == on and n > 100 and np.array_equal(m12, om)
    # windows: each row sees a random third of the other side, independently per direction
    def windows(n1, n2):
        off, idx = [0], []
        for i in range(n1):
            idx.extend(np.sort(rng.choice(n2, n2 // 3, replace=False)).tolist() if i % 11 else [])
            off.append(len(idx))
        return np.array(off, np.int32), np.array(idx, np.int32)
    c12, c21 = windows(nA, nB), windows(nB, nA)
    n, m12 = ctx.match_mutual(A, B, 0.8, c12, c21)
    on, om = orc.mutual_nn(A, B, 0.8, c12, c21)
    assert n == on and 5 < n < 150 and np.array_equal(m12, om)


# ---- reduced-precision convolution modes (precision 1: bf16 hi/lo split, 2: plain bf16) ----

@pytest.mark.parametrize("precision", [1, 2, 3])
def test_reduced_precision_cnn_vs_oracle(precision, pkg, onet, synth):
    """The bf16 MFMA kernels against the oracle's statement of the same arithmetic (operands rounded to bf16 hi
    (+ lo), exact products, fp32 accumulation).  Sizes cover the 8 x 32 and 16 x 16 tiles, ragged edges and the
    fused first layer.  Tolerance: the two sides accumulate in different orders, so from the second layer on
    their fp32 inputs differ by ~1e-6 relative, which is enough to flip the rounding of an operand's last kept
    bit: 2^-17 (8e-6) of the value with the hi/lo split, 2^-8 (4e-3) with plain bf16.  Measured: 0.4-1.6e-5
    (heat 2-3e-5) and 1.5-3.6e-3 (heat 7e-3) of the tensor's scale; a misplaced operand half shows as >= 1e-3
    in the split mode."""
    tol = {1: (4e-5, 1e-4), 2: (1.5e-2, 3e-2), 3: (1e-5, 3e-5)}[precision]
    for (H, W, seed, kind) in ((72, 104, 41, "sparse"), (128, 160, 42, "dense"), (240, 320, 43, "sparse")):
        img = synth.make_frame(H, W, seed, 0, kind)
        ctx = get_ctx(pkg, W, H, precision=precision)
        ctx.inference([img])
        ctx.detect_keypoints(1)
        ctx.detect_lines(1)
        t = onet.inference(img, precision)
        for which, k in ((pkg.T_FEAT, "feat"), (pkg.T_JUNC, "junc"), (pkg.T_HEAT_RAW, "heat"), (pkg.T_DESCMAP, "desc")):
            got, ref = ctx.read_tensor(which), nhwc(t[k])
            err, scale = float(np.abs(got - ref).max()), float(np.abs(ref).max())
            assert err <= tol[k == "heat"] * scale, (H, W, k, err, scale)


def _by_coord(r):
    return {tuple(int(v) for v in p): i for i, p in enumerate(r["xy"])}


@pytest.mark.parametrize("name,kind", FULL)
def test_bf16_split_full_frame_vs_fp32_golden(name, kind, pkg, synth):
    """precision 1 against the fp32 goldens of the reference nets: descriptors within the 1e-3 bar at every key
    point both lists hold, key-point sets equal except where scores are within the split's error (2e-4) of a
    competitor or of the cap, the edge sets nearly equal (a few heat pixels sit within 1e-3 of a threshold)."""
    g = golden(name)
    H, W, seed, maxnum = [int(v) for v in g["meta"]]
    ctx = get_ctx(pkg, W, H, maxnum, precision=1)
    for t in (0, 1):
        r = ctx.extract(synth.make_frame(H, W, seed, t, kind))
        assert float(np.abs(ctx.read_tensor(pkg.T_SCORE)[::4, ::4] - g[f"score_sub_{t}"]).max()) < 3e-4
        a, b = _by_coord(r), {tuple(int(v) for v in p): i for i, p in enumerate(g[f"kp_xy_{t}"])}
        common = sorted(set(a) & set(b))
        assert len(common) >= 0.98 * len(b), (len(common), len(b))
        ia, ib = [a[p] for p in common], [b[p] for p in common]
        assert float(np.linalg.norm(r["desc"][ia] - g[f"kp_desc_{t}"][ib], axis=1).max()) < 1e-3
        assert float(np.abs(r["score"][ia] - g[f"kp_score_{t}"][ib]).max()) < 3e-4
        ea, eb = edge_coord_set(r["edge_idx"], r["xy"]), edge_coord_set(g[f"edge_idx_{t}"], g[f"kp_xy_{t}"])
        assert len(ea ^ eb) <= max(4, 0.05 * len(eb)), (len(ea ^ eb), len(eb))


def test_plain_bf16_deviation_report(pkg, synth, capsys):
    """precision 2 is outside the parity bar by construction; this records how far (and that the path runs)."""
    g = golden(FULL[0][0])
    H, W, seed, maxnum = [int(v) for v in g["meta"]]
    ctx = get_ctx(pkg, W, H, maxnum, precision=2)
    r = ctx.extract(synth.make_frame(H, W, seed, 0, FULL[0][1]))
    a, b = _by_coord(r), {tuple(int(v) for v in p): i for i, p in enumerate(g["kp_xy_0"])}
    common = sorted(set(a) & set(b))
    dd = np.linalg.norm(r["desc"][[a[p] for p in common]] - g["kp_desc_0"][[b[p] for p in common]], axis=1)
    with capsys.disabled():
        print(f"\nplain bf16 vs fp32: {len(common)}/{len(b)} key points in common, descriptor L2 max {dd.max():.3e} "
              f"mean {dd.mean():.3e}")
    assert len(common) >= 0.8 * len(b)
    assert float(dd.max()) < 0.5


@pytest.mark.parametrize("precision", [1, 2, 3])
def test_reduced_precision_batch_equals_single(precision, pkg, synth):
    """The reduced-precision kernels are deterministic and independent of the position in the batch: every frame of
    a batch is bit-identical to the same frame extracted alone (same kernels, same order of operations)."""
    H, W = 96, 128
    imgs = [synth.make_frame(H, W, 50 + k, 0, "sparse" if k % 2 else "dense") for k in range(4)]
    one = get_ctx(pkg, W, H, precision=precision)
    singles = [one.extract(i) for i in imgs]
    ctx = get_ctx(pkg, W, H, batch=4, precision=precision)
    for r, s in zip(ctx.extract_batch(imgs), singles):
        for k in ("xy", "score", "un", "out", "desc", "edge_idx", "conn_idx", "coline_pairs"):
            assert np.array_equal(r[k], s[k], equal_nan=True), k


@pytest.mark.parametrize("name,kind", FULL)
def test_f16_split_full_frame_vs_fp32_golden(name, kind, pkg, synth):
    """precision 3 against the fp32 goldens of the reference nets, with the tolerances of the fp32 path's own
    end-to-end test doubled: score map 2e-5, key points equivalent up to ties within 4e-5, descriptors 1e-4
    (bar 1e-3), edge sets within two edges."""
    g = golden(name)
    H, W, seed, maxnum = [int(v) for v in g["meta"]]
    ctx = get_ctx(pkg, W, H, maxnum, precision=3)
    for t in (0, 1):
        r = ctx.extract(synth.make_frame(H, W, seed, t, kind))
        assert float(np.abs(ctx.read_tensor(pkg.T_SCORE)[::4, ::4] - g[f"score_sub_{t}"]).max()) < 2e-5
        swaps = assert_keypoints_equivalent(r["xy"], r["score"], g[f"kp_xy_{t}"], g[f"kp_score_{t}"], tol=4e-5)
        ea, eb = edge_coord_set(r["edge_idx"], r["xy"]), edge_coord_set(g[f"edge_idx_{t}"], g[f"kp_xy_{t}"])
        assert len(ea ^ eb) <= 2, len(ea ^ eb)
        if swaps == 0:
            assert float(np.linalg.norm(r["desc"] - g[f"kp_desc_{t}"], axis=1).max()) < 1e-4


def test_f16_split_reports_saturation(pkg, synth, tmp_path):
    """precision 3 saturates activations at +-2047 and must say so: with conv1a's weights scaled 40x the first layer's
    outputs leave that range and the library fails with PPG_ERANGE; the untouched weights pass."""
    import struct
    blob = bytearray(open(pkg.DEFAULT_WEIGHTS, "rb").read())
    assert blob[:8] == b"PPGW0001"
    n = struct.unpack_from("<I", blob, 8)[0]
    hit = False
    for k in range(n):
        base = 16 + 96 * k
        name = bytes(blob[base:base + 56]).split(b"\0")[0].decode()
        off, nbytes = struct.unpack_from("<QQ", blob, base + 80)  # name[56], ndim, dims[4], 4 bytes of padding
        if name in ("backbone.conv1a.weight", "backbone.conv1a.bias"):
            w = np.frombuffer(bytes(blob[off:off + nbytes]), np.float32) * np.float32(40.0)
            blob[off:off + nbytes] = w.tobytes()
            hit = True
    assert hit
    path = str(tmp_path / "scaled.bin")
    open(path, "wb").write(bytes(blob))
    img = synth.make_frame(96, 128, 3, 0, "dense")
    good = pkg.Context(pkg.default_config(128, 96, precision=3))
    assert good.extract(img)["n"] > 0
    good.close()
    # the warm-up frame of ppg_create already runs the networks, so the error may surface there
    with pytest.raises(pkg.PPGError, match="range of the fp16 split"):
        ctx = pkg.Context(pkg.default_config(128, 96, weights=path, precision=3))
        ctx.extract(img)


def test_two_cnn_lanes_equal_one(pkg, synth, monkeypatch):
    """PPG_CNN_LANES=2 (consecutive batches alternate between two CNN streams with their own intermediates):
    pipelined results must equal the single-lane blocking results bit for bit."""
    H, W = 96, 128
    seqs = [[synth.make_frame(H, W, s, t, k) for s, k in ((2, "sparse"), (9, "dense"))] for t in range(5)]
    one = get_ctx(pkg, W, H, 500, batch=2)
    ref = [one.extract_batch(seqs[t]) for t in range(5)]
    monkeypatch.setenv("PPG_CNN_LANES", "2")
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=2, max_num=500))
    monkeypatch.delenv("PPG_CNN_LANES")
    got = []
    for t in range(5):
        ctx.submit_batch(seqs[t], match_prev=False)
        if t >= 2:
            got.append(ctx.wait_batch())
    while len(got) < 5:
        got.append(ctx.wait_batch())
    for t in range(5):
        frames = got[t][0]
        for r, s in zip(frames, ref[t]):
            for k in ("xy", "score", "desc", "edge_idx", "conn_idx", "coline_pairs"):
                assert np.array_equal(r[k], s[k], equal_nan=True), (t, k)
    ctx.close()


def test_unfused_first_layer_equals_fused(pkg, synth, monkeypatch):
    """PPG_UNFUSED_CONV1A=1 runs conv1a as its own kernel (writing the full-resolution 64-channel tensor) instead of
    inside conv1b's staging: the same fused-multiply-add chain, so every output is bit-identical."""
    H, W = 72, 104
    img = synth.make_frame(H, W, 61, 0, "dense")
    fused = get_ctx(pkg, W, H).extract(img)
    feat_fused = get_ctx(pkg, W, H).read_tensor(pkg.T_FEAT).copy()
    monkeypatch.setenv("PPG_UNFUSED_CONV1A", "1")
    ctx = pkg.Context(pkg.default_config(W, H))
    monkeypatch.delenv("PPG_UNFUSED_CONV1A")
    r = ctx.extract(img)
    assert np.array_equal(ctx.read_tensor(pkg.T_FEAT), feat_fused)
    for k in ("xy", "score", "desc", "edge_idx", "conn_idx", "coline_pairs"):
        assert np.array_equal(r[k], fused[k], equal_nan=True), k
    ctx.close()


def test_device_line_tail_equals_host_statement(pkg, synth, monkeypatch, capsys):
    """The order-dependent tail of detectLines runs on the device (relaxation of the sequential overlap pass, score
    thresholds, compaction, collinearity); PPG_LINES_HOST=1 runs the host statement instead.  Both must give the same
    graph bit for bit, and the device must keep all but a few frames (it hands a frame to the host only when a libm
    result sits at a float rounding boundary or a work array overflows)."""
    H, W = 480, 640
    imgs = [synth.make_frame(H, W, 800 + k, k % 3, "dense" if k % 2 else "sparse") for k in range(8)]
    dev = get_ctx(pkg, W, H, 1024, batch=8)
    f0, h0 = dev.tail_stats()
    got = dev.extract_batch(imgs)
    f1, h1 = dev.tail_stats()
    monkeypatch.setenv("PPG_LINES_HOST", "1")
    host = pkg.Context(pkg.default_config(W, H, max_batch=8, max_num=1024))
    monkeypatch.delenv("PPG_LINES_HOST")
    ref = host.extract_batch(imgs)
    assert host.tail_stats() == (16, 8)   # the warm-up batch has no key points; the 8 real frames ran on the host
    host.close()
    for r, s in zip(got, ref):
        assert r["n_edges"] == s["n_edges"] > 20
        for k in ("edge_idx", "edge_score", "conn_off", "conn_idx", "coline_off", "coline_pairs"):
            assert np.array_equal(r[k], s[k], equal_nan=True), k
    with capsys.disabled():
        print(f"\ndevice line tail: {h1 - h0} of {f1 - f0} frames handed to the host")
    assert f1 - f0 == 8 and h1 - h0 <= 2


def test_fused_match_equals_matrix_path(pkg, orc, synth, monkeypatch):
    """Resident all-pairs SearchForInitialization: the fused path (MFMA prefilter, exact chain for the listed
    columns, claim pass as a fixed point) against the path through the full distance matrix (PPG_MATCH_MATRIX=1)
    and against the oracle, on real frames including few-key-point ones, bit for bit."""
    H, W = 480, 640
    kinds = [("dense", 1024), ("sparse", 1024)]
    for kind, maxn in kinds:
        seqs = [[synth.make_frame(H, W, 900 + s, t, kind) for s in range(3)] for t in range(3)]
        # a nearly empty stream: fewer than 10 key points -> all-zero descriptors -> every distance is 0
        blank = np.full((H, W), 128, np.uint8)
        blank[100:140, 200:260] = 255
        for t in range(3):
            seqs[t].append(blank)
        fused = get_ctx(pkg, W, H, maxn, batch=4)
        res = []
        for t in range(3):
            frames = fused.extract_batch(seqs[t])
            res.append((frames, fused.match_prev_batch(4, 0.9) if t else None))
        monkeypatch.setenv("PPG_MATCH_MATRIX", "1")
        ref = pkg.Context(pkg.default_config(W, H, max_batch=4, max_num=maxn))
        monkeypatch.delenv("PPG_MATCH_MATRIX")
        cfg = orc.default_cfg(W, H, maxn)
        for t in range(3):
            ref.extract_batch(seqs[t])
            if not t:
                continue
            cnts_r, m_r = ref.match_prev_batch(4, 0.9)
            cnts_f, m_f = res[t][1]
            assert np.array_equal(cnts_f, cnts_r) and np.array_equal(m_f, m_r), (kind, t)
            for b in (0, 3):
                f0, f1 = res[t - 1][0][b], res[t][0][b]
                co, mo, _ = orc.search_for_initialization(cfg, 0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], 0)
                assert cnts_f[b] == co and np.array_equal(m_f[b, :f0["n"]], mo), (kind, t, b)
            assert cnts_f[:3].min() > 50
        ref.close()


def test_fused_match_adversarial_descriptors(pkg, monkeypatch):
    """All-pairs SearchForInitialization through ppg_match_nn on descriptor sets built to break the matrix-free path:
    tight clusters (the prefilter lists overflow, the extended lists of the claim pass do not fit), exact duplicates
    (ties at distance 0 and between rows), all-zero rows, very few columns, more rows than columns.  Bit-equal to the
    path through the full distance matrix (PPG_MATCH_MATRIX=1) in matches, best and second-best distances."""
    rng = np.random.default_rng(5)

    def unit(x):
        n = np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
        x = (x / np.maximum(n, np.float32(1e-20))).astype(np.float32)
        # keep the squared norms at or below 1 (the fused path's precondition, which the library checks on host input)
        return (x * np.float32(0.99999)).astype(np.float32)

    def clustered(n, k, eps):
        centres = unit(rng.normal(size=(k, 256)).astype(np.float32))
        return unit(centres[rng.integers(0, k, n)] + eps * rng.normal(size=(n, 256)).astype(np.float32))

    cases = []
    cases.append(("clusters", clustered(1000, 6, 2e-3), clustered(1024, 6, 2e-3)))
    cases.append(("tight clusters", clustered(900, 3, 1e-5), clustered(950, 3, 1e-5)))
    base = unit(rng.normal(size=(40, 256)).astype(np.float32))
    cases.append(("duplicates", base[rng.integers(0, 40, 800)], base[rng.integers(0, 40, 1024)]))
    z = unit(rng.normal(size=(600, 256)).astype(np.float32))
    z[::3] = 0.0
    zb = unit(rng.normal(size=(700, 256)).astype(np.float32))
    zb[::5] = 0.0
    cases.append(("zero rows", z, zb))
    cases.append(("three columns", unit(rng.normal(size=(700, 256)).astype(np.float32)), unit(rng.normal(size=(3, 256)).astype(np.float32))))
    cases.append(("one column", unit(rng.normal(size=(64, 256)).astype(np.float32)), unit(rng.normal(size=(1, 256)).astype(np.float32))))
    cases.append(("few rows", unit(rng.normal(size=(5, 256)).astype(np.float32)), unit(rng.normal(size=(900, 256)).astype(np.float32))))
    noisy = unit(rng.normal(size=(1024, 256)).astype(np.float32))
    cases.append(("same set", noisy, unit(noisy + 0.05 * rng.normal(size=noisy.shape).astype(np.float32))))
    fused = pkg.Context(pkg.default_config(128, 96, max_batch=1, max_num=500))
    monkeypatch.setenv("PPG_MATCH_MATRIX", "1")
    ref = pkg.Context(pkg.default_config(128, 96, max_batch=1, max_num=500))
    monkeypatch.delenv("PPG_MATCH_MATRIX")
    try:
        for name, A, B in cases:
            for th, ratio in ((0.7, 0.9), (2.5, 1.0)):
                mf, bf, sf, cf = fused.match_nn(A, B, th=th, ratio=ratio, mode=pkg.MATCH_INIT)
                mr, br, sr, cr = ref.match_nn(A, B, th=th, ratio=ratio, mode=pkg.MATCH_INIT)
                assert cf == cr and np.array_equal(mf, mr), (name, th, ratio, int(np.sum(mf != mr)))
                assert np.array_equal(bf, br) and np.array_equal(sf, sr), (name, th, ratio)
    finally:
        fused.close()
        ref.close()


def test_waited_results_survive_the_next_submit(pkg, synth):
    """Four batches in flight, the bench's pattern: the buffers behind the frames that ppg_wait_batch hands out must
    still hold that batch after ONE more submit (the ring has one slot more than may be in flight)."""
    H, W = 96, 128
    ctx = get_ctx(pkg, W, H, 500, batch=4)
    seqs = [[synth.make_frame(H, W, 300 + 7 * t + s, t, "dense" if (s + t) % 2 else "sparse") for s in range(3)] for t in range(7)]
    ref = [ctx.extract_batch(seqs[t]) for t in range(7)]
    for t in range(4):
        ctx.submit_batch(seqs[t], match_prev=t > 0)
    for t in range(3):
        outs, _, _, n = ctx.wait_batch(to_dict=False)      # raw pointers into the slot of batch t
        ctx.submit_batch(seqs[4 + t], match_prev=True)      # the next submit must not touch them
        frames = [pkg._frame_to_dict(outs[i]) for i in range(n)]
        for r, s in zip(frames, ref[t]):
            for k in ("xy", "score", "un", "desc", "edge_idx", "edge_score", "conn_idx", "coline_pairs"):
                assert np.array_equal(r[k], s[k], equal_nan=True), (t, k)
    for t in range(4):
        ctx.wait_batch()


def test_extend_map_matches_device_walk(pkg, orc, synth, monkeypatch):
    """The point-pair-graph matcher as one wave on the device against the oracle and against the host walk
    (PPG_EXTEND_HOST=1), on dense frames with many map points and edges; then a graph whose degrees exceed the
    kernel's limits, which must fall back to the host walk and still agree (more listed map edges than the kernel holds);
    then wide windows (more than 64 key points each) over a graph whose seeds have weight matrices of more than 64
    entries; then few points with more than 64 edges each (the walk itself runs into the limit and hands over)."""
    H, W = 480, 640
    ctx = get_ctx(pkg, W, H, 1024)
    cfg = orc.default_cfg(W, H, 1024)
    r = ctx.extract(synth.make_frame(H, W, 11, 0, "dense"))
    monkeypatch.setenv("PPG_EXTEND_HOST", "1")
    host = pkg.Context(pkg.default_config(W, H, max_num=1024))
    monkeypatch.delenv("PPG_EXTEND_HOST")
    rng = np.random.default_rng(15)
    for trial, (m, nedges, th) in enumerate(((800, 1500, 4.0), (1500, 4000, 7.0), (300, 9000, 10.0), (300, 5000, 40.0), (150, 4000, 10.0))):
        g = random_map_graph(rng, r["desc"], r["un"], min(m, r["n"]), nedges)
        mm = g["desc"].shape[0]
        if trial == 4:   # inside the kernel's array sizes, beyond its edges per point
            assert g["edge_off"][-1] <= 16384 and np.diff(g["edge_off"]).max() > 64
        tracked = (rng.random(mm) < 0.05).astype(np.uint8)
        fmp = np.full(r["n"], -1, np.int32)
        pre = rng.choice(r["n"], size=40, replace=False)
        fmp[pre] = rng.integers(0, mm, 40)
        fme = np.full(r["n_edges"], -1, np.int32)
        args = (0.8, th, g, r["desc"], r["un"], r["edge_idx"], r["conn_off"], r["conn_idx"], tracked, fmp, fme)
        co, to, po, eo = orc.extend_map_matches(cfg, *args)
        cg, tg, pg, eg = ctx.extend_map_matches(*args)
        ch, th_, ph, eh = host.extend_map_matches(*args)
        assert cg == co == ch and np.array_equal(tg, to) and np.array_equal(pg, po) and np.array_equal(eg, eo), trial
        assert np.array_equal(th_, to) and np.array_equal(ph, po) and np.array_equal(eh, eo)
        assert co > 100 and (eo >= 0).sum() > 0
    # a map that repeats the frame's own graph: most points are reached by the seed growing, through long queues
    for th in (4.0, 12.0):
        g = random_map_graph(rng, r["desc"], r["un"], min(900, r["n"]), 300, r["edge_idx"])
        mm = g["desc"].shape[0]
        g["in_view"][:] = 1
        g["bad"][:] = 0
        tracked = np.zeros(mm, np.uint8)
        fmp = np.full(r["n"], -1, np.int32)
        fme = np.full(r["n_edges"], -1, np.int32)
        args = (0.8, th, g, r["desc"], r["un"], r["edge_idx"], r["conn_off"], r["conn_idx"], tracked, fmp, fme)
        co, to, po, eo = orc.extend_map_matches(cfg, *args)
        cg, tg, pg, eg = ctx.extend_map_matches(*args)
        ch, th_, ph, eh = host.extend_map_matches(*args)
        assert cg == co == ch and np.array_equal(tg, to) and np.array_equal(pg, po) and np.array_equal(eg, eo), th
        assert np.array_equal(th_, to) and np.array_equal(ph, po) and np.array_equal(eh, eo)
        grown = int(to.sum()) - co // 2   # tracked points that were not seeds
        print(f"graph-shaped map, th {th}: {int(to.sum())} of {mm} map points tracked, {grown} by seed growing, {int((eo >= 0).sum())} edges")
        assert grown > 100 and (eo >= 0).sum() > 100
    host.close()
