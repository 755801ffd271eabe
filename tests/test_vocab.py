"""DBoW3 vocabulary: the library's reader against the oracle's independent one (CPU), and the GPU tree descent plus
the BowVector / FeatureVector assembly against the oracle (GPU).  DBoW3 itself is absent: parity against it is unpinned."""
import os

import numpy as np
import pytest

REF_VOCS = ["/root/reference/Vocabulary/voc_euroc_9x3.gz", "/root/reference/Vocabulary/voc_tum_9x3.gz"]


def _same_tree(pkg_voc, o):
    assert (pkg_voc.k, pkg_voc.L, pkg_voc.n_nodes, pkg_voc.n_words, pkg_voc.dim, pkg_voc.scoring, pkg_voc.weighting) == \
        (o["k"], o["L"], o["n_nodes"], o["n_words"], o["dim"], o["scoring"], o["weighting"])
    for i in range(o["n_nodes"]):
        par, w, wid, ch, d = pkg_voc.node(i)
        assert par == o["parent"][i] and w == o["weight"][i] and wid == o["word_id"][i]
        assert ch == o["child_idx"][o["child_off"][i]:o["child_off"][i + 1]].tolist()
        assert np.array_equal(d, o["desc"][i])


def test_synthetic_vocabulary_roundtrip(pkg, tmp_path):
    from oracle import vocab as ov
    path = str(tmp_path / "voc.bin")
    ov.write_synthetic(path, k=5, L=2, dim=64, seed=3)
    o = ov.load(path)
    assert o["n_nodes"] == 1 + 5 + 25 and o["n_words"] == 25
    v = pkg.Vocabulary(path)
    _same_tree(v, o)
    v.close()
    with pytest.raises(pkg.PPGError, match="signature|short"):
        bad = tmp_path / "bad.bin"
        bad.write_bytes(b"\0" * 64)
        pkg.Vocabulary(str(bad))


@pytest.mark.parametrize("path", REF_VOCS)
def test_reference_vocabulary_files(path, pkg):
    """The shipped vocabularies (k = 9, L = 3, 820 nodes, 729 words, one QuickLZ-packed chunk at the end): the library's
    C++ reader and the oracle's Python reader agree on every node.  Runs where the reference is present."""
    if not os.path.exists(path):
        pytest.skip("the reference's Vocabulary/ files are not on this machine")
    from oracle import vocab as ov
    o = ov.load(path)
    assert (o["k"], o["L"], o["n_nodes"], o["n_words"], o["dim"]) == (9, 3, 820, 729, 256)
    assert o["packed_chunks"] >= 1                      # the QuickLZ decoder is exercised
    assert sorted(o["word_node"].tolist()) == sorted(i for i in range(820) if o["child_off"][i + 1] == o["child_off"][i])
    v = pkg.Vocabulary(path)
    _same_tree(v, o)
    # levelsup = 4 > L = 3 (map/src/Frame.cpp:131): every feature is filed under the root
    rng = np.random.default_rng(0)
    from oracle import oracle as orc
    desc = rng.normal(size=(50, 256)).astype(np.float32)
    desc /= np.linalg.norm(desc, axis=1, keepdims=True)
    w, wt, nd = orc.bow_transform(o, desc, 4)
    assert not nd.any() and (w >= 0).all() and len(set(w.tolist())) > 5
    bow, fv = v.bow_vectors(w, wt, nd)
    obow, ofv = ov.bow_vectors(o, w, wt, nd)
    assert [b[0] for b in bow] == [b[0] for b in obow] and np.allclose([b[1] for b in bow], [b[1] for b in obow], rtol=1e-15, atol=0)
    assert fv == ofv and len(fv) <= 1
    assert abs(sum(b[1] ** 2 for b in bow) - 1.0) < 1e-12    # L2 scoring: unit BowVector
    v.close()


@pytest.mark.gpu
def test_bow_transform_vs_oracle(pkg, orc, synth, tmp_path):
    """Tree descent on the GPU for real descriptors (host pointers and the resident batch) against the oracle, for
    levelsup inside the tree and above it; then the two containers."""
    from oracle import vocab as ov
    path = str(tmp_path / "voc.bin")
    ov.write_synthetic(path, k=9, L=3, dim=256, seed=7)
    o = ov.load(path)
    v = pkg.Vocabulary(path)
    H, W = 480, 640
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=2, max_num=500))
    try:
        imgs = [synth.make_frame(H, W, 5, 0, "sparse"), synth.make_frame(H, W, 6, 0, "dense")]
        fr = ctx.extract_batch(imgs)
        for levelsup in (4, 1, 2, 0):
            bw, bwt, bnd = pkg.bow_transform_batch(ctx, v, 2, levelsup)
            for b in range(2):
                n = fr[b]["n"]
                ow, owt, ond = orc.bow_transform(o, fr[b]["desc"], levelsup)
                assert np.array_equal(bw[b, :n], ow) and np.array_equal(bwt[b, :n], owt) and np.array_equal(bnd[b, :n], ond)
                assert (bw[b, n:] == -1).all()
                hw, hwt, hnd = pkg.bow_transform(ctx, v, fr[b]["desc"], levelsup)
                assert np.array_equal(hw, ow) and np.array_equal(hwt, owt) and np.array_equal(hnd, ond)
                bow, fv = v.bow_vectors(ow, owt, ond)
                obow, ofv = ov.bow_vectors(o, ow, owt, ond)
                assert [x[0] for x in bow] == [x[0] for x in obow]
                assert np.allclose([x[1] for x in bow], [x[1] for x in obow], rtol=1e-14, atol=0)
                assert fv == ofv
                if levelsup >= 3:
                    assert [q[0] for q in fv] == [0]
                else:
                    assert len(fv) > 3
            assert len(set(bw[1, :fr[1]["n"]].tolist())) > 50
        # a descriptor with NaN components is nearer to no child (DBoW3 itself would never return): the descent takes
        # the first child at every level and ends
        bad = fr[1]["desc"][:4].copy()
        bad[1, 7] = np.nan
        bad[3, :] = np.nan
        hw, hwt, hnd = pkg.bow_transform(ctx, v, bad, 1)
        ow, owt, ond = orc.bow_transform(o, bad, 1)
        assert np.array_equal(hw, ow) and np.array_equal(hwt, owt) and np.array_equal(hnd, ond)
        assert hw[1] == hw[3] and hw[1] >= 0
    finally:
        ctx.close()
        v.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,L", [(2, 5), (3, 4), (10, 2), (20, 2), (17, 1), (9, 3)])
def test_bow_transform_tree_shapes(k, L, pkg, orc, tmp_path):
    """Tree descent on the GPU against the oracle for other branching factors and depths than the shipped 9 x 3: more
    children than the sixteen lanes that share a node, a single level, deep narrow trees; every levelsup from the leaves
    to above the root; descriptors equal to a node's own (distance zero) and far from all."""
    from oracle import vocab as ov
    path = str(tmp_path / f"voc_{k}_{L}.bin")
    ov.write_synthetic(path, k=k, L=L, dim=256, seed=100 * k + L)
    o = ov.load(path)
    v = pkg.Vocabulary(path)
    ctx = pkg.Context(pkg.default_config(128, 96))
    try:
        rng = np.random.default_rng(k * 31 + L)
        d = rng.normal(size=(300, 256)).astype(np.float32)
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        nodes = np.asarray(o["desc"] if isinstance(o, dict) else o.desc, np.float32).reshape(-1, 256)
        take = rng.integers(1, nodes.shape[0], 60)
        d[:60] = nodes[take]                       # exactly a node's descriptor
        d[60:80] *= 40.0                           # far from everything
        for levelsup in range(0, L + 2):
            hw, hwt, hnd = pkg.bow_transform(ctx, v, d, levelsup)
            ow, owt, ond = orc.bow_transform(o, d, levelsup)
            assert np.array_equal(hw, ow) and np.array_equal(hwt, owt) and np.array_equal(hnd, ond), (k, L, levelsup)
        assert len(set(hw.tolist())) > min(k ** L, 300) // 8
    finally:
        ctx.close()
        v.close()
