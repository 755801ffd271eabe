"""Parity statistics of the reduced-precision modes on the benchmark's own workload: the 32 dense camera streams x 8
frames that bench.py replays (640 x 480, JUNCTION_MAX_NUM = 1024), every stream matched frame to frame.  Per mode (1: bf16
hi/lo split, 3: fp16 hi/lo split) against mode 0 (fp32, the parity path): the share of frames whose key-point list, edge
list and match vector are identical, and the largest descriptor distance at a common key point, which must stay inside
the 1e-3 bar.  Prints the table and writes it to gpurun_out/precision_stats.json."""
import json
import os

import numpy as np
import pytest

from conftest import edge_coord_set

pytestmark = pytest.mark.gpu

H, W, B, T, MAXN = 480, 640, 32, 8, 1024


def _run_mode(pkg, host, precision):
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=MAXN, precision=precision))
    frames, matches = [], []
    try:
        for t in range(T):
            ctx.submit_batch(list(host[t]), match_prev=t > 0, nnratio=0.9)
            fr, cnts, m12, n = ctx.wait_batch(to_dict=True, want_matches=True)
            assert n == B
            frames.append(fr)
            matches.append((cnts.copy(), m12.copy()))
    finally:
        ctx.close()
    return frames, matches


def test_fast_modes_against_fp32_on_the_bench_streams(pkg, synth, capsys):
    host = np.empty((T, B, H, W), np.uint8)
    for s in range(B):
        scene = synth.make_scene(H, W, 1000 * s + 11, "dense")
        for t in range(T):
            host[t, s] = synth.frame_from_scene(scene, H, W, t)
    ref_frames, ref_matches = _run_mode(pkg, host, 0)
    table = {}
    for mode, name in ((1, "bf16 hi/lo split"), (3, "fp16 hi/lo split")):
        frames, matches = _run_mode(pkg, host, mode)
        st = dict(frames=0, kp_identical=0, kp_same_set=0, edges_identical=0, matches_identical=0, pairs=0,
                  kp_total=0, kp_common=0, max_desc_l2=0.0, max_score_diff=0.0, match_entries=0, match_entries_equal=0,
                  edge_sets_equal=0, edges_total=0, edges_common=0, match_sets_equal=0, matches_total=0, matches_common=0)
        for t in range(T):
            for b in range(B):
                r, f = ref_frames[t][b], frames[t][b]
                st["frames"] += 1
                same_list = r["n"] == f["n"] and np.array_equal(r["xy"], f["xy"])
                st["kp_identical"] += same_list
                rs = {tuple(p): i for i, p in enumerate(r["xy"].tolist())}
                fs = {tuple(p): i for i, p in enumerate(f["xy"].tolist())}
                st["kp_same_set"] += set(rs) == set(fs)
                common = [(rs[p], fs[p]) for p in rs if p in fs]
                st["kp_total"] += r["n"]
                st["kp_common"] += len(common)
                if common:
                    ia, ib = np.array(common).T
                    d = np.linalg.norm(r["desc"][ia].astype(np.float64) - f["desc"][ib].astype(np.float64), axis=1)
                    st["max_desc_l2"] = max(st["max_desc_l2"], float(d.max()))
                    st["max_score_diff"] = max(st["max_score_diff"], float(np.abs(r["score"][ia] - f["score"][ib]).max()))
                st["edges_identical"] += same_list and np.array_equal(r["edge_idx"], f["edge_idx"])
                # independent of the numbering of the key points: edges as pairs of end-point coordinates
                er, ef = edge_coord_set(r["edge_idx"], r["xy"]), edge_coord_set(f["edge_idx"], f["xy"])
                st["edge_sets_equal"] += er == ef
                st["edges_total"] += len(er)
                st["edges_common"] += len(er & ef)
                if t > 0:
                    st["pairs"] += 1
                    prev_same = ref_frames[t - 1][b]["n"] == frames[t - 1][b]["n"] and \
                        np.array_equal(ref_frames[t - 1][b]["xy"], frames[t - 1][b]["xy"])
                    mr, mf = ref_matches[t][1][b], matches[t][1][b]
                    st["matches_identical"] += bool(prev_same and same_list and np.array_equal(mr, mf))
                    if prev_same and same_list:
                        n1 = ref_frames[t - 1][b]["n"]
                        st["match_entries"] += n1
                        st["match_entries_equal"] += int((mr[:n1] == mf[:n1]).sum())
                    # matches as (position in the previous frame, position in this frame)
                    def pairs(prev, cur, m12):
                        return {(tuple(prev["xy"][i].tolist()), tuple(cur["xy"][j].tolist())) for i, j in enumerate(m12[:prev["n"]].tolist()) if j >= 0}
                    pr, pf = pairs(ref_frames[t - 1][b], r, mr), pairs(frames[t - 1][b], f, mf)
                    st["match_sets_equal"] += pr == pf
                    st["matches_total"] += len(pr)
                    st["matches_common"] += len(pr & pf)
        table[name] = st
        # the bar of the metric: descriptors within 1e-3 of the fp32 path at every common key point
        assert st["max_desc_l2"] <= 1e-3, (name, st["max_desc_l2"])
        assert st["kp_common"] >= 0.98 * st["kp_total"], (name, st["kp_common"], st["kp_total"])
    lines = ["frames with identical ...  |  shares over all frames",
             "mode              frames  kp lists  kp sets  edge lists  edge sets(xy)  match vectors  match sets(xy) |  common kp  common edges  common matches  max desc L2  max score diff"]
    for name, st in table.items():
        lines.append("%-17s %6d  %7.1f%%  %6.1f%%  %9.1f%%  %12.1f%%  %12.1f%%  %13.1f%% |  %8.3f%%  %11.3f%%  %13.3f%%  %11.2e  %14.2e" % (
            name, st["frames"], 100.0 * st["kp_identical"] / st["frames"], 100.0 * st["kp_same_set"] / st["frames"],
            100.0 * st["edges_identical"] / st["frames"], 100.0 * st["edge_sets_equal"] / st["frames"],
            100.0 * st["matches_identical"] / max(st["pairs"], 1), 100.0 * st["match_sets_equal"] / max(st["pairs"], 1),
            100.0 * st["kp_common"] / max(st["kp_total"], 1), 100.0 * st["edges_common"] / max(st["edges_total"], 1),
            100.0 * st["matches_common"] / max(st["matches_total"], 1), st["max_desc_l2"], st["max_score_diff"]))
    with capsys.disabled():
        print("\n" + "\n".join(lines))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "precision_stats.json"), "w") as fh:
            json.dump(table, fh, indent=1)
