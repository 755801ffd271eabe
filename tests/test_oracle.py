"""CPU tests: the oracle against the golden fixtures (outputs of the reference's TorchScript nets run
with torch fp32 on CPU, tools/make_goldens.py) and against hand-checkable cases of the reference's
post-processing and matcher semantics."""
import zlib

import numpy as np
import pytest

from conftest import golden, assert_keypoints_equivalent, edge_coord_set

SMALL = ["cnn_96x128_s1_sparse.npz", "cnn_128x160_s7_dense.npz", "cnn_64x96_s3_sparse.npz"]
FULL = [("full_480x640_s1_sparse.npz", "sparse"), ("full_480x640_s11_dense.npz", "dense"), ("full_512x512_s5_sparse.npz", "sparse")]


@pytest.mark.parametrize("name", SMALL)
def test_cnn_matches_reference_nets(name, orc, onet):
    g = golden(name)
    t = onet.inference(g["img"])
    # fp32 with a different summation order than oneDNN: relative 2e-6 of the tensor scale
    # (the heat logits pass three batch norms with scales up to ~10, hence the wider band)
    for k, rel in (("feat", 4e-6), ("junc", 4e-6), ("heat", 1e-5), ("desc", 4e-6)):
        err, scale = float(np.abs(t[k] - g[k]).max()), float(np.abs(g[k]).max())
        assert err <= rel * scale, (k, err, scale)
    assert np.abs(orc.score_map(t["junc"]) - g["score"]).max() < 1e-5
    assert np.abs(orc.heat_prob(t["heat"]) - g["prob"]).max() < 2e-5


@pytest.mark.parametrize("name", SMALL)
def test_post_processing_on_reference_tensors(name, orc):
    """Oracle post-processing fed the golden torch tensors reproduces the stored integer outputs
    (regression pin of the restatement) and torch's grid_sampler + normalize descriptors."""
    g = golden(name)
    H, W = g["img"].shape
    cfg = orc.default_cfg(W, H)
    kp = orc.detect_keypoints(cfg, g["score"])
    assert np.array_equal(kp["xy"], g["kp_xy"])
    assert np.array_equal(kp["score"], g["kp_score"])
    refined = orc.refine_heat(cfg, g["prob"])
    assert np.array_equal(refined, g["refined"])
    ln = orc.detect_lines(cfg, refined, kp["un"], kp["out"])
    assert np.array_equal(ln["edge_idx"], g["edge_idx"])
    assert np.array_equal(ln["conn_idx"], g["conn_idx"])
    assert np.array_equal(ln["coline_pairs"], g["coline_pairs"])
    d = orc.gen_descriptors(cfg, g["desc"], kp["xy"])
    # K11/K12 against torch.grid_sampler + F.normalize: 1e-3 is the required descriptor tolerance, we are ~1e-6
    assert np.linalg.norm(d - g["kp_desc"], axis=1).max() < 1e-5


@pytest.mark.parametrize("name,kind", FULL)
def test_full_frame_end_to_end(name, kind, orc, onet, synth):
    g = golden(name)
    H, W, seed, maxnum = [int(v) for v in g["meta"]]
    cfg = orc.default_cfg(W, H, maxnum)
    img = synth.make_frame(H, W, seed, 0, kind)
    assert np.uint32(zlib.crc32(img.tobytes())) == g["img_crc_0"]
    r = orc.extract(onet, cfg, img)
    assert np.abs(r["score_map"][::4, ::4] - g["score_sub_0"]).max() < 1e-5
    # refineHeatMap is discontinuous at the 0.01 cut, so compare by the share of pixels that moved
    assert np.mean(np.abs(r["heat"][::4, ::4] - g["refined_sub_0"]) > 1e-4) < 1e-3
    # Integer outputs against (torch CNN -> oracle post-processing).  The two CNN evaluations differ by
    # fp32 summation order (~2e-6 in score), so near-tied key points may swap places; everything else
    # must agree.  The sparse scenes agree exactly.
    swaps = assert_keypoints_equivalent(r["xy"], r["score"], g["kp_xy_0"], g["kp_score_0"])
    ea, eb = edge_coord_set(r["edge_idx"], r["xy"]), edge_coord_set(g["edge_idx_0"], g["kp_xy_0"])
    assert len(ea ^ eb) <= 2, len(ea ^ eb)
    if kind == "sparse":
        assert swaps == 0 and np.array_equal(r["edge_idx"], g["edge_idx_0"])
        assert np.linalg.norm(r["desc"] - g["kp_desc_0"], axis=1).max() < 1e-4


@pytest.mark.parametrize("name,kind", FULL[:1])
def test_search_for_initialization_golden(name, kind, orc):
    g = golden(name)
    H, W, seed, maxnum = [int(v) for v in g["meta"]]
    cfg = orc.default_cfg(W, H, maxnum)
    for win, tag in ((50, "w50"), (0, "all")):
        cnt, m12, prev = orc.search_for_initialization(cfg, 0.9, g["kp_desc_0"], g["kp_desc_1"], g["kp_un_1"], g["kp_un_0"], win)
        assert cnt == int(g[f"nmatch_{tag}"])
        assert np.array_equal(m12, g[f"m12_{tag}"])
        # the scene moves by (-3, -1) px per frame: accepted matches must agree with it
        ok = m12 >= 0
        d = g["kp_un_1"][m12[ok]] - g["kp_un_0"][ok]
        assert np.mean(np.all(np.abs(d - np.array([-3, -1])) <= 4, axis=1)) > 0.9


def test_empty_and_flat_images(orc, onet):
    cfg = orc.default_cfg(96, 64)
    r = orc.extract(onet, cfg, np.full((64, 96), 128, np.uint8))
    assert r["n"] == 0 and "desc" not in r  # detectLines/genPointDescriptor never reached usable points


def test_nms_semantics(orc):
    W, H = 64, 48
    cfg = orc.default_cfg(W, H)
    s = np.zeros((H, W), np.float32)
    s[10, 10] = 0.9
    s[12, 13] = 0.8      # within 4 px of (10,10): suppressed
    s[10, 15] = 0.7      # 5 px away: kept
    s[2, 30] = 0.99      # border (y < 4): rejected without suppressing anything
    s[5, 30] = 0.5       # kept although (30,2) is stronger
    s[20, 20] = 1.0 / 128.0  # exactly at threshold: kept (comparison is `<`)
    s[30, 30] = np.float32(1.0 / 128.0) - np.float32(1e-6)
    kp = orc.detect_keypoints(cfg, s)
    assert [tuple(p) for p in kp["xy"]] == [(10, 10), (15, 10), (30, 5), (20, 20)]
    cfg.junction_max_num = 2
    kp = orc.detect_keypoints(cfg, s)
    assert [tuple(p) for p in kp["xy"]] == [(10, 10), (15, 10)]


def test_refine_heat_rules(orc):
    W, H = 32, 16
    cfg = orc.default_cfg(W, H)
    h = np.zeros((H, W), np.float32)
    h[:, :16] = 0.5            # saturated tile: >= 90 % valid and percentile > 0.1 -> zeroed
    h[0:4, 16:32] = np.linspace(0.02, 0.4, 64, dtype=np.float32).reshape(4, 16)
    r = orc.refine_heat(cfg, h)
    assert np.all(r[:, :16] == 0)
    vals = np.sort(h[0:4, 16:32].ravel())[::-1]
    ave = np.float32(np.sum(vals[:int(np.float32(0.3) * 64)].astype(np.float64)) / np.float32(int(np.float32(0.3) * 64)))
    exp = np.minimum(h[0:4, 16:32] / ave, 1).astype(np.float32)
    assert np.array_equal(r[0:4, 16:32], exp)
    assert np.all(r[4:, 16:] == 0)


def test_short_edge_nan_quirk(orc):
    """segNum == 1 (5 <= dist < 6 px) gives 0/0 scores that pass both `<` tests (PPGExtractor.cpp:497,512)."""
    W, H = 64, 48
    cfg = orc.default_cfg(W, H)
    heat = np.ones((H, W), np.float32)
    un = np.array([[10, 10], [15, 10]], np.float32)
    ln = orc.detect_lines(cfg, heat, un, np.zeros(2, np.uint8))
    assert ln["n_edges"] == 1 and np.isnan(ln["edge_score"][0])


def test_grid_window_order(orc):
    cfg = orc.default_cfg(640, 480)
    pos = np.array([[100, 100], [105, 100], [100, 130], [639, 100], [90, 95]], np.float32)
    idx = orc.features_in_area(cfg, pos, 100.0, 100.0, 50.0)
    # x-cell outer, y-cell inner, insertion order; (639,100) rounds to cell 64 and is never stored
    assert list(idx) == [4, 0, 2, 1]
    assert len(orc.features_in_area(cfg, pos, 639.0, 100.0, 5.0)) == 0


def test_extend_map_matches_small_graph(orc):
    rng = np.random.default_rng(0)
    cfg = orc.default_cfg(640, 480)
    n = 6
    desc = rng.normal(size=(n, 256)).astype(np.float32)
    desc /= np.linalg.norm(desc, axis=1, keepdims=True)
    pos = np.array([[100, 100], [150, 100], [200, 100], [100, 200], [300, 300], [400, 100]], np.float32)
    # key edges: 0-1, 1-2, 0-3
    edge_idx = np.array([[0, 1], [1, 2], [0, 3]], np.uint32)
    conn = [[0, 2], [0, 1], [1], [2], [], []]
    conn_off = np.cumsum([0] + [len(c) for c in conn]).astype(np.int32)
    conn_idx = np.array([e for c in conn for e in c], np.uint32)
    # map points 0..3 mirror key points 0..3 (noisy descriptors), map edges mirror the key edges
    m = 4
    mdesc = desc[:m] + 0.01 * rng.normal(size=(m, 256)).astype(np.float32)
    mdesc /= np.linalg.norm(mdesc, axis=1, keepdims=True)
    medges = [(0, 1), (1, 2), (0, 3)]
    inc = [[] for _ in range(m)]
    for eid, (a, b) in enumerate(medges):
        inc[a].append((eid, b))
        inc[b].append((eid, a))
    g = dict(desc=mdesc.astype(np.float32), proj=pos[:m].copy(), viewcos=np.ones(m, np.float32),
             in_view=np.array([1, 0, 0, 0], np.uint8),  # only point 0 is searched directly; the rest must come from the graph
             bad=np.zeros(m, np.uint8), nobs=np.ones(m, np.int32),
             edge_off=np.cumsum([0] + [len(i) for i in inc]).astype(np.int32),
             edge_id=np.array([e for i in inc for e, _ in i], np.int32),
             edge_other=np.array([o for i in inc for _, o in i], np.int32),
             edge_ok=np.ones(sum(len(i) for i in inc), np.uint8))
    cnt, tracked, fmp, fme = orc.extend_map_matches(cfg, 0.8, 3.0, g, desc, pos, edge_idx, conn_off, conn_idx,
                                                    np.zeros(m, np.uint8), np.full(n, -1, np.int32), np.full(3, -1, np.int32))
    # seed 0 -> key 0; its map edges (0-1),(0-3) pair with key edges 0,2.  Key 1 is then popped, but the map
    # side stays map point 0 (Matcher.cpp:296), so 1-2 is never followed: map point 2 stays unmatched.
    assert list(fmp[:4]) == [0, 1, -1, 3]
    assert list(fme) == [0, -1, 2]
    assert cnt == 2 and list(tracked) == [1, 1, 0, 1]


def _tiny_map(nobs0):
    """Three map points at depth 1 in front of an identity camera (fx = fy = 100, c = (32, 24))."""
    uv = np.array([[20.0, 20.0], [22.0, 21.0], [50.0, 40.0]], np.float32)
    pos = np.stack([(uv[:, 0] - 32) / 100, (uv[:, 1] - 24) / 100, np.ones(3)], 1).astype(np.float32)
    desc = np.zeros((3, 256), np.float32)
    desc[0, 0] = 1.0
    desc[1, 0], desc[1, 1] = 0.8, 0.6      # 0.632 from e0, 0.894 from e1
    desc[2, 2] = 1.0
    nrm = pos / np.linalg.norm(pos, axis=1, keepdims=True)
    d = np.linalg.norm(pos, axis=1).astype(np.float32)
    return uv, {"pos": pos, "normal": nrm.astype(np.float32), "min_dist": 0.5 * d, "max_dist": 2 * d, "desc": desc,
                "bad": np.zeros(3, np.uint8), "nobs": np.array([nobs0, 0, 1], np.int32)}


def _tiny_cfg(orc):
    cfg = orc.default_cfg(64, 48)
    for k, v in enumerate([100, 0, 32, 0, 100, 24, 0, 0, 1]):
        cfg.K[k] = float(v)
    return cfg


def test_search_by_projection_claim_and_overwrite(orc):
    """Matcher.cpp:31-87: a slot taken by an observed map point is skipped by later rows (:67-69); a slot taken by an
    unobserved one is overwritten (:80)."""
    cfg = _tiny_cfg(orc)
    cur_pos = np.array([[21.0, 20.0], [23.0, 22.0]], np.float32)
    cur_desc = np.zeros((2, 256), np.float32)
    cur_desc[0, 0] = 1.0
    cur_desc[1, 1] = 1.0
    R, t = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    last_mp, outl = np.array([0, 1, 2], np.int32), np.zeros(3, np.uint8)
    _, mp = _tiny_map(1)
    n, cur = orc.search_by_projection_frames(cfg, R, t, mp, last_mp, outl, cur_pos, cur_desc, np.full(2, -1, np.int32), 15.0)
    assert n == 1 and cur.tolist() == [0, -1]          # row 1 finds slot 0 blocked and slot 1 too far (0.894 > 0.8)
    _, mp = _tiny_map(0)
    n, cur = orc.search_by_projection_frames(cfg, R, t, mp, last_mp, outl, cur_pos, cur_desc, np.full(2, -1, np.int32), 15.0)
    assert n == 2 and cur.tolist() == [1, -1]          # row 1 overwrites row 0's unobserved point
    # outliers and points behind the camera are skipped
    n, cur = orc.search_by_projection_frames(cfg, R, t, mp, last_mp, np.array([1, 0, 0], np.uint8), cur_pos, cur_desc,
                                             np.full(2, -1, np.int32), 15.0)
    assert n == 1 and cur.tolist() == [1, -1]
    n, cur = orc.search_by_projection_frames(cfg, R, np.array([0, 0, -2], np.float32), mp, last_mp, outl, cur_pos, cur_desc,
                                             np.full(2, -1, np.int32), 15.0)
    assert n == 0


def test_sim3_projection_and_fuse_gates(orc):
    """Matcher.cpp:479-568 and :897-1135: already-found points and matched slots are skipped, the scale and viewing
    angle gates hold, Fuse's reprojection gate (e^2 <= 5.99) applies only to the first variant."""
    cfg = _tiny_cfg(orc)
    uv, mp = _tiny_map(1)
    kf_pos = np.array([[21.0, 20.0], [23.0, 22.0], [50.0, 43.0]], np.float32)
    kf_desc = np.zeros((3, 256), np.float32)
    kf_desc[0, 0] = 1.0
    kf_desc[1, 1] = 1.0
    kf_desc[2, 2] = 1.0
    R, t = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    pts = np.array([0, 1, 2], np.int32)
    n, m = orc.search_by_projection_sim3(cfg, R, t, mp, pts, kf_pos, kf_desc, np.full(3, -1, np.int32), 10, 1.0)
    assert n == 2 and m.tolist() == [0, -1, 2]           # point 1: slot 0 taken, slot 1 too far (0.894 > 0.7)
    # point 0 already found in another slot: skipped; slot 0 stays free for point 1 (0.632 <= 0.7)
    n, m = orc.search_by_projection_sim3(cfg, R, t, mp, pts, kf_pos, kf_desc, np.array([-1, 0, -1], np.int32), 10, 1.0)
    assert m.tolist() == [1, 0, 2] and n == 2
    # Fuse: point 2 projects 3 px from its key point: inside the window, outside e^2 <= 5.99
    n, b = orc.fuse_match(cfg, R, t, mp, pts, np.zeros(3, np.uint8), kf_pos, kf_desc, 4.0, True)
    assert b.tolist() == [0, 0, -1] and n == 2
    n, b = orc.fuse_match(cfg, R, t, mp, pts, np.zeros(3, np.uint8), kf_pos, kf_desc, 4.0, False)
    assert b.tolist() == [0, 0, 2] and n == 3
    # IsInKeyFrame, nullptr, scale gate
    mp2 = dict(mp)
    mp2["min_dist"] = mp["min_dist"].copy()
    mp2["min_dist"][1] = 5.0
    n, b = orc.fuse_match(cfg, R, t, mp2, np.array([0, 1, -1], np.int32), np.array([1, 0, 0], np.uint8), kf_pos, kf_desc, 4.0, False)
    assert b.tolist() == [-1, -1, -1] and n == 0


def test_distinctive_descriptor_known_answer(orc):
    """feature/src/MapPoint.cpp:268-296: the row with the least median distance; BestMedian starts at 1.0, so
    when every median is >= 1 row 0 stays."""
    e = np.eye(256, dtype=np.float32)
    # four observations along a line and one far away; median = third smallest of the sorted row (N = 5, own 0 included):
    # row 0 -> 0.057, rows 1 and 2 -> 0.028 (the first of them wins), row 3 -> 0.057
    d = np.stack([e[0], e[0] * 0.98 + e[1] * 0.02, e[0] * 0.96 + e[1] * 0.04, e[0] * 0.94 + e[1] * 0.06, e[5]])
    assert orc.distinctive_descriptor(d) == 1
    assert orc.distinctive_descriptor(np.stack([e[0], e[1], e[2]])) == 0      # all medians sqrt(2) >= 1
    assert orc.distinctive_descriptor(e[:1]) == 0
    assert orc.distinctive_descriptor(np.stack([e[3], e[3]])) == 0            # N = 2: median = vDists[0] = 0 < 1


def test_undistort_maps_and_remap(orc):
    """PPGExtractor.cpp:65-71, 261-262 restated after OpenCV: without distortion the maps are the pixel grid and the
    remap is the identity; with distortion the corners pull inwards/outwards and a linear ramp stays a ramp."""
    cfg = orc.default_cfg(64, 48)
    mx, my = orc.undistort_maps(cfg)
    jj, ii = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(48, dtype=np.float32))
    assert np.abs(mx - jj).max() < 1e-4 and np.abs(my - ii).max() < 1e-4
    rng = np.random.default_rng(3)
    img = rng.random((48, 64), dtype=np.float32)
    assert np.array_equal(orc.remap_heat(cfg, img), img)
    cfg.D[0], cfg.D[1], cfg.D[2], cfg.D[3] = -0.28, 0.07, 2e-4, 2e-5
    mx, my = orc.undistort_maps(cfg)
    cx, cy = cfg.K[2], cfg.K[5]
    # barrel distortion (k1 < 0): the source of a destination pixel lies closer to the centre
    assert abs(mx[0, 0] - cx) < abs(0 - cx) and abs(my[0, 0] - cy) < abs(0 - cy)
    assert abs(mx[24, 32] - 32) < 0.05 and abs(my[24, 32] - 24) < 0.05
    ramp = (0.25 * jj + 0.5 * ii).astype(np.float32)
    out = orc.remap_heat(cfg, ramp)
    inside = (mx > 0) & (mx < 62) & (my > 0) & (my < 46)
    want = 0.25 * np.round(mx * 32) / 32 + 0.5 * np.round(my * 32) / 32
    assert np.abs(out - want)[inside].max() < 1e-4
    # fisheye maps: radially symmetric about the principal point for fx = fy
    cfg.fisheye = 1
    cfg.K[0] = cfg.K[4] = 50.0
    cfg.D[0], cfg.D[1], cfg.D[2], cfg.D[3] = 0.01, -0.002, 0.0, 0.0
    mx, my = orc.undistort_maps(cfg)
    assert abs((mx[24, 40] - 32) - (my[32, 32] - 24)) < 1e-4


def test_feature_vector_matchers_small(orc):
    """Matcher.cpp:393-477 / 663-754 / 767-885 on two nodes: only features of the same node are compared (lock-step
    walk with lower_bound), a matched frame slot is not offered again, the ratio test is strict."""
    cfg = _tiny_cfg(orc)
    e = np.eye(256, dtype=np.float32)
    d1 = np.stack([e[0], 0.8 * e[0] + 0.6 * e[1], e[2], e[3]])       # KF features 0..3
    d2 = np.stack([e[0], e[2], 0.6 * e[0] + 0.8 * e[1], e[3]])       # F features 0..3
    mp1 = np.array([0, 1, 2, -1], np.int32)
    bad = np.zeros(3, np.uint8)
    fv1 = [(1, [0, 1]), (4, [2]), (9, [3])]
    fv2 = [(1, [0, 2]), (3, [3]), (4, [1])]
    n, out = orc.search_by_bow_frame(cfg, 0.9, fv1, fv2, d1, mp1, bad, d2)
    # KF0 -> F0 (0 < 0.9 * 0.894); KF1: F0 taken, F2 at 0.283 with no second (1e6) -> F2; KF2 -> F1 in node 4; KF3 has no MP
    assert n == 3 and out.tolist() == [0, 2, 1, -1]                  # indexed by frame slot: slot 1 <- point 2, slot 2 <- point 1
    n, out = orc.search_by_bow_frame(cfg, 0.9, [(2, [0, 1, 2, 3])], fv2, d1, mp1, bad, d2)
    assert n == 0 and out.tolist() == [-1, -1, -1, -1]               # no common node
    bad[1] = 1
    n, out = orc.search_by_bow_frame(cfg, 0.9, fv1, fv2, d1, mp1, bad, d2)
    assert n == 2 and out.tolist() == [0, 2, -1, -1]                 # slot 2 stays free: its only suitor is bad
    # key frame / key frame: KF2 slots need a good map point of their own; results are KF2's map points
    bad[:] = 0
    mp2 = np.array([2, -1, 0, 1], np.int32)
    n, out = orc.search_by_bow_keyframes(cfg, 0.9, fv1, fv2, d1, mp1, d2, mp2, bad)
    assert n == 2 and out.tolist() == [2, 0, -1, -1]                 # KF1 feature 2: its only candidate has no map point
    # triangulation: only features WITHOUT map points pair up; a distant epipole and a pure x translation
    pos1 = np.array([[10, 10], [20, 12], [30, 30], [40, 20]], np.float32)
    pos2 = pos1 + np.array([2.0, 0.0], np.float32)
    d = np.stack([e[0], e[1], e[2], e[3]])
    none = np.full(4, -1, np.int32)
    root = [(0, [0, 1, 2, 3])]
    R, t2 = np.eye(3, dtype=np.float32), np.array([0.05, 0.0, 0.0], np.float32)
    n, pairs = orc.search_for_triangulation(cfg, R, np.zeros(3, np.float32), R, t2, root, root, d, pos1, none, d, pos2, none)
    assert n == 4 and pairs.tolist() == [[0, 0], [1, 1], [2, 2], [3, 3]]
    n, pairs = orc.search_for_triangulation(cfg, R, np.zeros(3, np.float32), R, t2, root, root, d, pos1,
                                            np.array([-1, 5, -1, -1], np.int32), d, pos2, np.array([-1, -1, 7, -1], np.int32))
    assert n == 2 and pairs.tolist() == [[0, 0], [3, 3]]
    # off the epipolar line (a vertical offset of 5 px with a horizontal baseline): rejected
    pos2b = pos1 + np.array([2.0, 5.0], np.float32)
    n, pairs = orc.search_for_triangulation(cfg, R, np.zeros(3, np.float32), R, t2, root, root, d, pos1, none, d, pos2b, none)
    assert n == 0


def test_aten_op_sequence_matches_goldens_and_oracle(onet, synth):
    """The plain PyTorch (ATen CPU) op sequence over the extracted weights reproduces the reference archives' outputs
    (committed goldens) and agrees with the C restatement of the CNN."""
    from oracle.aten_cnn import AtenNets
    from conftest import golden
    nets = AtenNets()
    g = golden("cnn_96x128_s1_sparse.npz")
    img = g["img"]
    feat, junc, heat, desc = nets.inference(img)
    for name, t in (("feat", feat), ("junc", junc), ("heat", heat), ("desc", desc)):
        ref = g[name]
        got = t[0].numpy()
        assert got.shape == ref.shape
        assert float(np.abs(got - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max())), name
    o = onet.inference(img)
    assert float(np.abs(o["feat"] - feat[0].numpy()).max()) <= 1e-5 * float(np.abs(o["feat"]).max())


def test_reduced_precision_statements_against_the_reference_nets(onet):
    """The oracle's statements of the build's reduced-precision convolution modes, measured against the goldens of the
    reference nets: the fp16 hi/lo split stays at the fp32 noise floor, the bf16 split within a few 1e-5, plain bf16 at
    the percent level (what each mode of the library is then compared with on the GPU)."""
    g = golden("cnn_96x128_s1_sparse.npz")
    bound = {0: 4e-6, 3: 8e-6, 1: 1e-4, 2: 5e-2}
    floor = {2: 1e-3}
    for prec, tol in bound.items():
        t = onet.inference(g["img"], prec)
        worst = max(float(np.abs(t[k] - g[k]).max() / np.abs(g[k]).max()) for k in ("feat", "junc", "desc"))
        assert worst <= tol, (prec, worst)
        assert worst >= floor.get(prec, 0.0), (prec, worst)


def test_line_graph_sensitivity_to_libm_reading(orc, onet, synth, capsys):
    """detectLines evaluates atan2, abs and sin on floats (feature/src/PPGExtractor.cpp:283, :325, :330, :413).  The literal
    reading is the float overloads of the host's libm (ppgo_cfg.math_mode 0: :283 and :413 are qualified std:: calls, the
    unqualified ones resolve to the float overloads under libstdc++'s C wrappers); oracle, library and GPU tail default to
    a chosen definition instead, the double functions rounded to float (mode 1), which does not depend on the libm.  This
    measures, on the golden images and on full-size frames of the benchmark's dense streams, that no output of the line
    graph - edges, scores, per-point adjacency, collinear pairs - differs between the two.  Mode 2 (integer abs, a
    different algorithm) is reported only."""
    from conftest import golden
    cases = []
    for name in ("cnn_96x128_s1_sparse.npz", "cnn_128x160_s7_dense.npz", "cnn_64x96_s3_sparse.npz"):
        g = golden(name)
        H, W = g["img"].shape
        cases.append((name, W, H, 500, g["refined"], g["kp_un"], g["kp_out"]))
    # seeds 1000 * s + 11 are the scenes of bench.py's camera streams s; t is the frame of the stream
    for (seed, kind, maxn, t) in ((11, "dense", 1024, 0), (1, "sparse", 500, 0), (3011, "dense", 1024, 5), (17011, "dense", 1024, 2),
                                  (31011, "dense", 1024, 7)):
        H, W = 480, 640
        cfg = orc.default_cfg(W, H, maxn)
        t = onet.inference(synth.make_frame(H, W, seed, t, kind))
        kp = orc.detect_keypoints(cfg, orc.score_map(t["junc"]))
        cases.append((f"{kind}{seed}", W, H, maxn, orc.refine_heat(cfg, orc.heat_prob(t["heat"])), kp["un"], kp["out"]))
    report = []
    for name, W, H, maxn, refined, un, out in cases:
        res = []
        for mode in (0, 1, 2):
            cfg = orc.default_cfg(W, H, maxn)
            cfg.math_mode = mode
            res.append(orc.detect_lines(cfg, refined, un, out))
        for k in ("edge_idx", "conn_off", "conn_idx", "coline_off", "coline_pairs"):
            assert np.array_equal(res[0][k], res[1][k]), (name, k)
        assert np.array_equal(res[0]["edge_score"], res[1]["edge_score"], equal_nan=True)
        a = {tuple(e) for e in res[1]["edge_idx"].tolist()}
        b = {tuple(e) for e in res[2]["edge_idx"].tolist()}
        report.append(f"{name}: {len(a)} edges, float-vs-double libm reading: identical; integer abs changes {len(a ^ b)} edges")
    with capsys.disabled():
        print("\n" + "\n".join(report))


def _distances_in_order(a, B, packet, fused):
    """fp32 `(a - b).norm()` of one row against every row of B with the squares added in a given order: `packet` = 0 is
    the sequential chain d = 0..255; 4, 8 and 16 are Eigen's vectorised reduction for SSE, AVX and AVX-512 packets (two
    packet accumulators over alternating packets, their sum, then the halving horizontal add; Eigen/src/Core/Redux.h and
    the architectures' predux).  `fused`: the square goes into the sum without a rounding of its own (a compiler that
    contracts to FMA; the product of two floats is exact in double)."""
    f32, f64 = np.float32, np.float64
    d = (B - a[None, :]).astype(f32)

    def add_sq(p, x):
        if fused:
            return (p.astype(f64) + x.astype(f64) * x.astype(f64)).astype(f32)
        return (p + (x * x).astype(f32)).astype(f32)

    if packet == 0:
        s = np.zeros(d.shape[0], f32)
        for k in range(d.shape[1]):
            s = add_sq(s, d[:, k])
        return np.sqrt(s).astype(f32)
    P = packet
    p0, p1 = (d[:, 0:P] * d[:, 0:P]).astype(f32), (d[:, P:2 * P] * d[:, P:2 * P]).astype(f32)
    for i in range(2 * P, d.shape[1], 2 * P):
        p0, p1 = add_sq(p0, d[:, i:i + P]), add_sq(p1, d[:, i + P:i + 2 * P])
    p = (p0 + p1).astype(f32)
    while p.shape[1] > 4:
        h = p.shape[1] // 2
        p = (p[:, :h] + p[:, h:]).astype(f32)
    s = ((p[:, 0] + p[:, 2]).astype(f32) + (p[:, 1] + p[:, 3]).astype(f32)).astype(f32)
    return np.sqrt(s).astype(f32)


def _all_pairs_claim_match(D, th, ratio):
    """Matcher::SearchForInitialization (matching/src/Matcher.cpp:582-651) with a window that holds every feature, on a
    given distance matrix: rows in order, best and second over the columns not yet claimed (an accepted distance is
    stored in a vector<int>, :588, so a claimed column is never looked at again), accepted iff best <= th and
    best < ratio * second."""
    n1, n2 = D.shape
    claimed = np.zeros(n2, bool)
    m = np.full(n1, -1, np.int32)
    for i in range(n1):
        row = np.where(claimed, np.float32(np.inf), D[i])
        j = int(np.argmin(row))
        best = row[j]
        row[j] = np.inf
        if best <= np.float32(th) and best < np.float32(ratio) * row.min():
            m[i] = j
            claimed[j] = True
    return m


@pytest.mark.parametrize("name", ["full_480x640_s1_sparse.npz", "full_480x640_s11_dense.npz"])
def test_match_indices_do_not_depend_on_the_distance_summation_order(name, orc, capsys):
    """DescriptorDistance is `(des1 - des2).norm()` on Eigen maps (feature/src/MapPoint.cpp:22-29): the order in which Eigen
    adds the 256 squares depends on the instruction set it was compiled for and cannot be pinned without Eigen.  Oracle,
    library and kernels define it as the sequential fused chain.  This measures what the choice can change: the all-pairs
    claim match of the two golden frames is evaluated on distance matrices built in every order Eigen can use (SSE, AVX and
    AVX-512 packets, with and without contraction to FMA), the plain sequential sum and the correctly rounded one; the
    sequential fused chain must reproduce the stored matches of the oracle, and every other order the same index vector."""
    g = golden(name)
    A, B = g["kp_desc_0"], g["kp_desc_1"]
    orders = [("sequential, fused (the definition)", 0, True), ("sequential", 0, False), ("SSE packets", 4, False),
              ("SSE packets, fused", 4, True), ("AVX packets", 8, False), ("AVX packets, fused", 8, True),
              ("AVX-512 packets", 16, False)]
    mats = []
    for label, packet, fused in orders:
        mats.append((label, np.stack([_distances_in_order(A[i], B, packet, fused) for i in range(A.shape[0])])))
    exact = np.sqrt(((A[:, None, :].astype(np.float64) - B[None, :, :].astype(np.float64)) ** 2).sum(2)).astype(np.float32)
    mats.append(("correctly rounded", exact))
    ref = mats[0][1]
    # the emulated definition is the oracle's own chain, bit for bit
    for i in (0, 7, A.shape[0] - 1):
        for j in (0, 3, B.shape[0] - 1):
            assert ref[i, j] == np.float32(orc.descriptor_distance(A[i], B[j]))
    m_ref = _all_pairs_claim_match(ref, 0.7, 0.9)
    assert np.array_equal(m_ref, g["m12_all"])
    report = []
    for label, D in mats[1:]:
        m = _all_pairs_claim_match(D, 0.7, 0.9)
        report.append(f"{name}: {label}: max |difference of a distance| {float(np.abs(D - ref).max()):.2e}, "
                      f"{int((m != m_ref).sum())} of {len(m)} match indices differ")
        assert np.array_equal(m, m_ref), label
    with capsys.disabled():
        print("\n" + "\n".join(report))


def _greedy_nms(score, maxn, tie):
    """detectKeyPoint's selection (feature/src/PPGExtractor.cpp:168-205) written a second time, in numpy, with the order of
    equal scores as a parameter: "asc" = row-major index ascending (the definition of oracle and kernels), "desc", or the
    seed of a random order.  std::sort leaves that order open."""
    H, W = score.shape
    ys, xs = np.nonzero(score >= np.float32(1.0 / 128))
    s = score[ys, xs]
    idx = ys.astype(np.int64) * W + xs
    key = idx if tie == "asc" else -idx if tie == "desc" else np.random.default_rng(tie).permutation(len(idx))
    flag = np.zeros((H, W), bool)
    out = []
    for o in np.lexsort((key, -s.astype(np.float64))):
        x, y = int(xs[o]), int(ys[o])
        if x < 4 or x > W - 5 or y < 4 or y > H - 5 or flag[y, x]:
            continue
        out.append((x, y))
        if len(out) >= maxn:
            break
        flag[max(0, y - 4):y + 5, max(0, x - 4):x + 5] = True
    return out, int((np.sort(s)[1:] == np.sort(s)[:-1]).sum())


def test_keypoint_list_does_not_depend_on_the_sort_tie_order(orc, onet, synth, capsys):
    """The candidates of detectKeyPoint are ordered by std::sort on the score alone (feature/src/PPGExtractor.cpp:180): the
    order of equal scores is the C++ library's business.  Oracle and kernels define it (row-major index ascending).  On
    full-size frames of the benchmark's streams a few of the 20 000 candidates do carry equal scores; this runs a second,
    independent statement of the greedy selection with the ties ascending (must equal the oracle's list), descending and in
    three random orders, and requires the same key-point list every time."""
    report = []
    for seed, kind, t, maxn in ((11, "dense", 0, 1024), (1, "sparse", 0, 500), (3011, "dense", 5, 1024), (17011, "dense", 2, 1024)):
        score = orc.score_map(onet.inference(synth.make_frame(480, 640, seed, t, kind))["junc"])
        kp = orc.detect_keypoints(orc.default_cfg(640, 480, maxn), score)
        ref, n_ties = _greedy_nms(score, maxn, "asc")
        assert ref == [tuple(p) for p in kp["xy"].tolist()]
        for tie in ("desc", 1, 2, 3):
            assert _greedy_nms(score, maxn, tie)[0] == ref, (seed, tie)
        report.append(f"{kind}{seed}: {n_ties} pairs of equal scores among the candidates, {len(ref)} key points, "
                      "the same list under five tie orders")
    with capsys.disabled():
        print("\n" + "\n".join(report))
