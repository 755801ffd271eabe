#!/usr/bin/env python3
"""Front-end throughput benchmark: frames/sec of extract + describe + match at 640x480.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One *step* = one pass of the whole front-end over one batch of independent camera streams
(`--batch` frames, default 16, all resident in HBM before the timed region): the CNN stack, score map
+ NMS, line heat map + refine + line graph, descriptor sampling, and the all-pairs
SearchForInitialization match of every stream's frame against its previous frame.  Streams are
sharded over ranks (one process per GPU, weak scaling: every rank runs `--batch` streams); the only
collective is the global-shutter barrier — an all-reduce(MIN) of the per-rank frame sequence number
after every step.  Precision is fp32 operands with fp32 MFMA accumulation, the parity path.

Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel measured
with HIP events over the timed region, and a CPU baseline (the oracle port timed on this host).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2516.6  # v_mfma_f32_32x32x16_bf16: 32768 FLOP per 32 clk per SIMD x 1024 SIMDs x 2.4 GHz
PREROLL = 24                   # untimed steps guaranteed before the timed region (see main)


def conv_macs(H, W):
    """MACs per frame of every convolution, and of the ones run by conv_mfma_kernel<3,2,8,false>
    (conv1b runs in the <3,2,8,true> instantiation that also evaluates conv1a)."""
    h2, w2, h4, w4, h8, w8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
    layers = [  # (name, pixels, cin, cout, taps, dominant kernel?)
        ("conv1a", H * W, 1, 64, 9, False), ("conv1b", H * W, 64, 64, 9, False),
        ("conv2a", h2 * w2, 64, 64, 9, True), ("conv2b", h2 * w2, 64, 64, 9, True),
        ("conv3a", h4 * w4, 64, 128, 9, True), ("conv3b", h4 * w4, 128, 128, 9, True),
        ("conv4a", h8 * w8, 128, 128, 9, True), ("conv4b", h8 * w8, 128, 128, 9, True),
        ("convPa", h8 * w8, 128, 256, 9, True), ("convPb", h8 * w8, 256, 65, 1, False),
        ("convDa", h8 * w8, 128, 256, 9, True), ("convDb", h8 * w8, 256, 256, 1, False),
        ("heat0", h8 * w8, 128, 256, 9, True), ("heat1", h4 * w4, 64, 64, 9, True),
        ("heat2", h2 * w2, 16, 16, 9, False), ("heat3", H * W, 4, 2, 1, False)]
    total = sum(p * ci * co * t for _, p, ci, co, t, _ in layers)
    dom = sum(p * ci * co * t for _, p, ci, co, t, d in layers if d)
    ndom = sum(1 for l in layers if l[5])
    return total, dom, ndom


def cpu_baseline(H, W, kind, max_num, seconds=12.0):
    """The oracle (CPU port of the reference path) on this host: extract + match, frames/sec."""
    from oracle import oracle as orc
    synth = importlib.import_module("ppg_slam_amd.synth")
    net = orc.Net()
    cfg = orc.default_cfg(W, H, max_num)
    scene = synth.make_scene(H, W, 11, kind)
    prev = orc.extract(net, cfg, synth.frame_from_scene(scene, H, W, 0))
    n, t0 = 0, time.perf_counter()
    while True:
        cur = orc.extract(net, cfg, synth.frame_from_scene(scene, H, W, n + 1))
        if prev["n"] and cur["n"]:
            orc.search_for_initialization(cfg, 0.9, prev["desc"], cur["desc"], cur["un"], prev["un"], 0)
        prev = cur
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds or n >= 64:
            break
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    out = {"value": round(n / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{n} consecutive {W}x{H} '{kind}' frames, extract + all-pairs match vs previous frame; "
                     f"conv stack OpenMP over {cores} threads, post-processing and matcher single-threaded"}
    # What the reference's LibTorch CPU path spends in its four networks: the same ATen op sequence (conv2d, max_pool2d,
    # batch_norm, pixel_shuffle) over the extracted weights, torch CPU, same thread count.  CNN only; a bounded sample.
    try:
        import torch
        from oracle.aten_cnn import AtenNets
        torch.set_num_threads(cores)
        nets = AtenNets()
        img = synth.frame_from_scene(scene, H, W, 0)
        nets.inference(img)
        k, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 5.0 and k < 64:
            nets.inference(synth.frame_from_scene(scene, H, W, k + 1))
            k += 1
        out["aten_cpu_cnn_only"] = {"value": round(k / (time.perf_counter() - t1), 3), "unit": "frames/s", "cores": cores,
                                    "sample": f"{k} frames, the four networks only (torch {torch.__version__} CPU ops)"}
    except Exception as e:  # the baseline is a report, never a reason to fail the bench
        out["aten_cpu_cnn_only"] = {"error": str(e)[:200]}
    return out


SWEEP_BATCHES = (1, 2, 4, 8, 16)


def batch_sweep(pkg, synth, H, W, args, device_id):
    """Frames/s of the pipelined step at small batches (host images, upload included) and the latency of blocking
    single-frame calls."""
    out = {"note": "host images in every row (upload included); pipelined = ppg_submit_batch / ppg_wait_batch four deep with "
                   "match against the previous frame; blocking = one call at a time, results on the host before the next"}
    scene = [synth.make_scene(H, W, 1000 * s + 11, args.kind) for s in range(16)]
    T = 4
    imgs = [[synth.frame_from_scene(scene[s], H, W, t) for s in range(16)] for t in range(T)]
    rows = []
    for B in SWEEP_BATCHES:
        ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=args.max_num, device_id=device_id,
                                             precision=args.precision))
        def run(first, count):
            done = first
            for k in range(first, first + count):
                ctx.submit_batch(imgs[k % T][:B], match_prev=k > 0, nnratio=0.9)
                if k - done + 1 >= 4:
                    ctx.wait_batch(to_dict=False, want_matches=True)
                    done += 1
            while done < first + count:
                ctx.wait_batch(to_dict=False, want_matches=True)
                done += 1
        # at least 0.6 s of warm-up: right after the 32-stream passes the chip still runs at the clock its power limit
        # allowed them, and a one-frame step measured within the first 0.2 s is 12 % slower than in a process of its own
        warm = max(8, 256 // B)
        steps = max(16, 512 // B)
        tw = time.perf_counter()
        first = 0
        while True:
            run(first, warm)
            first += warm
            if time.perf_counter() - tw >= 0.6:
                break
        warm = first
        t0 = time.perf_counter()
        run(warm, steps)
        dt = time.perf_counter() - t0
        row = {"batch": B, "pipelined_frames_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4)}
        # blocking calls: ppg_extract_batch (upload, whole extract, results on the host), then the match against the
        # previous call's frames
        for k in range(6):
            ctx.extract_batch(imgs[k % T][:B], to_dict=False)
        nb = 40
        t0 = time.perf_counter()
        for k in range(nb):
            ctx.extract_batch(imgs[k % T][:B], to_dict=False)
        row["blocking_extract_ms"] = round((time.perf_counter() - t0) / nb * 1e3, 4)
        t0 = time.perf_counter()
        for k in range(nb):
            ctx.extract_batch(imgs[k % T][:B], to_dict=False)
            ctx.match_prev_batch(B, 0.9)
        row["blocking_extract_match_ms"] = round((time.perf_counter() - t0) / nb * 1e3, 4)
        rows.append(row)
        ctx.close()
    out["rows"] = rows
    out["single_frame_latency_ms"] = {"extract": rows[0]["blocking_extract_ms"],
                                      "extract_and_match": rows[0]["blocking_extract_match_ms"]}
    return out


def matcher_timings(pkg, synth, H, W, args, device_id):
    """Every matcher entry point once on a realistic problem (two consecutive dense frames with JUNCTION_MAX_NUM key
    points, a synthetic local map), wall time of the blocking C-ABI call including its uploads, with the CPU oracle's
    time for the same call beside it where the oracle restates it.  Milliseconds, best of 5."""
    from oracle import oracle as orc
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=2, max_num=args.max_num, device_id=device_id))
    cfg = orc.default_cfg(W, H, args.max_num)
    scene = synth.make_scene(H, W, 11, args.kind)
    f0, f1 = ctx.extract_batch([synth.frame_from_scene(scene, H, W, 0), synth.frame_from_scene(scene, H, W, 1)])
    rng = np.random.default_rng(1)

    def best(fn, reps=5):
        fn()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return round(min(ts) * 1e3, 4)

    out = {"problem": f"frames with {f0['n']} and {f1['n']} key points, {f1['n_edges']} key edges; local map of 600 points and 1500 edges",
           "note": "blocking calls with host pointers: uploads, kernels, downloads; cpu = the oracle's restatement on one core"}
    out["SearchForInitialization_window50"] = {
        "gpu": best(lambda: ctx.search_for_initialization(0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], 50)),
        "cpu": best(lambda: orc.search_for_initialization(cfg, 0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], 50), 2)}
    out["SearchForInitialization_all_pairs"] = {
        "gpu": best(lambda: ctx.search_for_initialization(0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], 0)),
        "gpu_resident": best(lambda: ctx.match_frames(0, 1, f0["n"], 0.9)),
        "cpu": best(lambda: orc.search_for_initialization(cfg, 0.9, f0["desc"], f1["desc"], f1["un"], f0["un"], 0), 2)}
    g = synth.random_map_graph(rng, f1["desc"], f1["un"], min(600, f1["n"]), 1500)
    m = g["desc"].shape[0]
    xargs = (0.8, 5.0, g, f1["desc"], f1["un"], f1["edge_idx"], f1["conn_off"], f1["conn_idx"], np.zeros(m, np.uint8),
             np.full(f1["n"], -1, np.int32), np.full(f1["n_edges"], -1, np.int32))
    out["ExtendMapMatches"] = {"gpu": best(lambda: ctx.extend_map_matches(*xargs)),
                               "cpu": best(lambda: orc.extend_map_matches(cfg, *xargs), 2),
                               "matches": int(ctx.extend_map_matches(*xargs)[0])}
    # the windowed argmin every projection matcher and Fuse run (ppg_match_projected's inner search), as candidate lists
    off, idx = [0], []
    for i in range(f0["n"]):
        idx.extend(orc.features_in_area(cfg, f1["un"], float(f0["un"][i, 0]), float(f0["un"][i, 1]), 15.0).tolist())
        off.append(len(idx))
    off, idx = np.array(off, np.int32), np.array(idx if idx else [0], np.int32)
    out["windowed_nearest_neighbour_r15"] = {"gpu": best(lambda: ctx.match_nn(f0["desc"], f1["desc"], off, idx, None, 0.8, 0.0, pkg.MATCH_PLAIN)),
                                             "candidates": int(off[-1])}
    # Matcher::SearchByProjection(Frame&, const Frame&, th) (Matcher.cpp:31-87), the matcher of every tracked frame: the
    # last frame's points projected into the current one search 15-pixel windows, accepted rows claim their slot
    uv = f0["un"] + np.array([3.0, 1.0], np.float32)
    out["SearchByProjection_r15"] = {
        "gpu": best(lambda: ctx.match_projected(f1["un"], f1["desc"], f0["desc"], uv, 15.0, 0.8, 0.0, None,
                                                np.ones(f0["n"], np.uint8), np.zeros(f1["n"], np.uint8))),
        "matches": int(ctx.match_projected(f1["un"], f1["desc"], f0["desc"], uv, 15.0, 0.8, 0.0, None,
                                           np.ones(f0["n"], np.uint8), np.zeros(f1["n"], np.uint8))[0])}
    segs = np.arange(0, f1["n"] + 1, 8, dtype=np.int32)
    out["ComputeDistinctiveDescriptors_8obs"] = {
        "gpu": best(lambda: ctx.distinctive_descriptors(f1["desc"][:segs[-1]], segs)),
        "cpu": best(lambda: [orc.distinctive_descriptor(f1["desc"][segs[p]:segs[p + 1]]) for p in range(len(segs) - 1)], 2),
        "map_points": int(len(segs) - 1)}
    import tempfile
    from oracle import vocab as ov
    with tempfile.TemporaryDirectory() as td:
        vp = os.path.join(td, "voc.bin")
        ov.write_synthetic(vp, k=9, L=3, dim=256, seed=7)
        o, v = ov.load(vp), pkg.Vocabulary(vp)
        out["BoW_transform_k9_L3"] = {"gpu": best(lambda: pkg.bow_transform(ctx, v, f1["desc"], 4)),
                                      "gpu_resident_2_frames": best(lambda: pkg.bow_transform_batch(ctx, v, 2, 4)),
                                      "cpu": best(lambda: orc.bow_transform(o, f1["desc"], 4), 2)}
        v.close()
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="independent camera streams per GPU per step")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--max-num", type=int, default=1024, help="JUNCTION_MAX_NUM (1024-keypoint match config)")
    ap.add_argument("--kind", default="dense")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-pass", action="store_true", help="skip the extra pass in the fp16 hi/lo-split mode")
    ap.add_argument("--precision", type=int, default=0,
                    help="0 fp32 (the parity path, the default and the number of record), 1 bf16 hi/lo split, 2 plain bf16, 3 fp16 hi/lo split")
    ap.add_argument("--host-input", action="store_true",
                    help="hand over host buffers in the MAIN timed region too (the default run reports that rate as with_upload)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep and the single-frame latency")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libppgfront has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    # PPG_BENCH_FORCE_DIST=1: set the process group up and run the barrier's collective in a world of one rank too
    # (rehearsal of the multi-GPU code path on a one-GPU box, under the launcher the multi-GPU runs use)
    force_dist = os.environ.get("PPG_BENCH_FORCE_DIST", "") not in ("", "0")
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist and "RANK" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = load_pkg()
    synth = importlib.import_module("ppg_slam_amd.synth")
    H, W, B = args.height, args.width, args.batch
    ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=args.max_num, device_id=local_rank,
                                         precision=args.precision))

    # B streams per rank; stream s of rank r is scene seed 1000*(r*B+s)+11; 8 consecutive frames of
    # every stream are staged in HBM and replayed round-robin.
    T = 8
    host = np.empty((T, B, H, W), np.uint8)
    for s in range(B):
        scene = synth.make_scene(H, W, 1000 * (rank * B + s) + 11, args.kind)
        for t in range(T):
            host[t, s] = synth.frame_from_scene(scene, H, W, t)
    frames = torch.from_numpy(host).cuda()
    streams_mod = importlib.import_module("ppg_slam_amd.streams")
    # The global-shutter barrier runs behind the library's C entry points (ppg_shutter_*: RCCL called from C on a stream
    # of the library's own); torch.distributed only carries the 128-byte communicator id to the ranks and the harness's
    # own barrier / max-over-ranks.  If the C entry points cannot be set up on every rank, all ranks fall back to the
    # same protocol over torch.distributed, and the line says so.
    shutter, shutter_kind = None, "none (one rank)"
    if dist is not None:
        try:
            uid = [None]
            if rank == 0:
                try:
                    uid[0] = streams_mod.shutter_unique_id()
                except Exception as e:   # noqa: BLE001
                    uid[0] = "error: " + str(e)[:200]
            dist.broadcast_object_list(uid, src=0)
            if not isinstance(uid[0], bytes):
                raise RuntimeError(str(uid[0]))
            ok = torch.ones(1, dtype=torch.int32, device="cuda")
            why = ""
            try:
                shutter = streams_mod.Shutter(ctx, rank, world, uid[0])
            except Exception as e:   # noqa: BLE001
                ok.zero_()
                why = str(e)[:200]
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) != 1:
                if shutter is not None:
                    shutter.close()
                shutter = None
                raise RuntimeError(why or "another rank could not create its shutter")
            shutter_kind = "ppg_shutter_* (C ABI, RCCL all-reduce(MIN) of an int32 on the library's stream)"
        except Exception as e:   # noqa: BLE001
            shutter_kind = "torch.distributed all_reduce(MIN) (fall-back: %s)" % str(e)[:200]
    if shutter is None:
        shutter = streams_mod.ShutterBarrier(dist, "cuda", single_rank_too=force_dist)

    class GroupNumbers:
        """The passes of this run restart their frame index; the barrier gets one ascending group number over all of them."""
        next = 0

        def release(self, _k):
            r = shutter.release(self.next)
            self.next += 1
            return r

    groups = GroupNumbers()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stats = {"kp": 0, "edges": 0, "matches": 0}

    do_match = [True]
    main_ctx = ctx

    def submit(k):
        mp = do_match[0] and k > 0
        if args.host_input:
            ctx.submit_batch(list(host[k % T]), match_prev=mp, nnratio=0.9)
        else:
            ctx.submit_batch_device(frames[k % T].data_ptr(), B, match_prev=mp, nnratio=0.9)

    def wait(k):
        outs, cnts, _, n = ctx.wait_batch(to_dict=False, want_matches=True)
        stats["kp"] += sum(outs[b].n for b in range(n))
        stats["edges"] += sum(outs[b].n_edges for b in range(n))
        stats["matches"] += int(cnts.sum())

    def run(first, count):
        """`count` steps, software-pipelined four deep (the CNN of batches k+3 and k+2, the post-processing of k+1 and
        the collection of k overlap); every collected group passes the global-shutter barrier (non-blocking all-reduce),
        and the barriers still in flight are waited for before the step count is complete."""
        streams_mod.run_steps(submit, wait, groups, first, count, depth=4)
        last = shutter.drain()
        # every rank runs the same steps: the newest group finished on all of them must be this rank's newest
        if count > 0 and last != groups.next - 1:
            raise RuntimeError("rank %d: the barrier released group %d, this rank finished %d: the ranks are out of step"
                               % (rank, last, groups.next - 1))

    # The first second of sustained load runs 5-10 % slower than the steady state (clock and power ramp of an
    # idle GPU, first-touch of the pinned result rings): an untimed pre-roll makes sure that at least PREROLL
    # steps have run before the timed region even when --warmup is small.
    preroll = max(0, PREROLL - args.warmup)
    run(0, preroll + args.warmup + 1)  # +1: the very first batch has no previous frame to match
    for key in stats:
        stats[key] = 0
    # Timed region: device events only around the dominant kernel (two per step), so that the measurement
    # does not perturb the throughput it reports.
    ctx.profile(2)
    barrier()
    t0 = time.perf_counter()
    run(preroll + args.warmup + 1, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    fl = ctx.last_profile_flops                # algorithmic FLOPs the library enqueued per kernel class
    ctx.profile(False)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    timed = dict(stats)                        # what the timed region saw
    # Second, untimed pass with events around every kernel class: the per-class breakdown.
    nb = min(args.steps, 12)
    ctx.profile(1)
    barrier()
    run(preroll + args.warmup + 1 + args.steps, nb)
    barrier()
    prof_all = ctx.profile_read()
    fl_all = ctx.last_profile_flops
    ctx.profile(False)
    # Third pass, extract only (no match against the previous frame): the second of the two rates the metric names.
    do_match[0] = False
    base = preroll + args.warmup + 1 + args.steps + nb
    run(base, 4)
    barrier()
    t1 = time.perf_counter()
    ne = max(nb, min(args.steps, 30))
    run(base + 4, ne)
    barrier()
    extract_only_fps = B * ne * world / (time.perf_counter() - t1)

    # The same step with the images handed over as host buffers (PPGExtractor::run takes a cv::Mat in host memory,
    # feature/src/PPGExtractor.cpp:151): staging into pinned memory and the upload are inside this timed region.
    with_upload = None
    if not args.host_input:
        do_match[0] = True
        args.host_input = True
        base2 = base + 4 + ne
        run(base2, 5)
        barrier()
        t3 = time.perf_counter()
        run(base2 + 5, args.steps)
        barrier()
        dt3 = time.perf_counter() - t3
        if dist is not None:
            tt = torch.tensor([dt3], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt3 = float(tt.item())
        args.host_input = False
        with_upload = {"value": round(B * args.steps * world / dt3, 1), "unit": "frames/s", "steps": args.steps,
                       "ms_per_step": round(dt3 / args.steps * 1e3, 4),
                       "what": "the same step with every image handed over in pageable host memory: the copy into pinned "
                               "staging memory and the host-to-device upload are inside the timed region"}

    # Batch-size sweep (rank 0's GPU only, single-GPU runs): the pipelined rate with B streams per step, and what one
    # blocking call per frame costs (the reference's own call pattern: one image per PPGExtractor::run).
    sweep = None
    if world == 1 and not args.no_sweep:
        sweep = batch_sweep(pkg, synth, H, W, args, local_rank)
        if args.precision == 0:
            # one camera per GPU in the two split modes (descriptors within 1e-3 of the fp32 path): the rows for one and two
            # frames per call
            global SWEEP_BATCHES
            keep = SWEEP_BATCHES
            SWEEP_BATCHES = (1, 2)
            try:
                for mode, key in ((1, "bf16_split_rows"), (3, "f16_split_rows")):
                    sub = argparse.Namespace(**vars(args))
                    sub.precision = mode
                    sweep[key] = batch_sweep(pkg, synth, H, W, sub, local_rank)["rows"]
            finally:
                SWEEP_BATCHES = keep
    matchers = None
    if world == 1 and not args.no_sweep and not args.no_cpu_baseline:
        try:
            matchers = matcher_timings(pkg, synth, H, W, args, local_rank)
        except Exception as e:   # a report beside the benchmark, never a reason to lose its line
            matchers = {"error": str(e)[:300]}

    # Fourth pass (fp32 runs only): the same step with the 3x3 convolutions in the fp16 hi/lo-split mode
    # (ppg_config.precision = 3), reported beside the number of record, never as it.
    split, split_bf16, plain_bf16 = None, None, None
    if args.precision == 0 and not args.no_split_pass:
        do_match[0] = True

        def split_pass(precision, dtype, what):
            nonlocal ctx
            ctx = pkg.Context(pkg.default_config(W, H, max_batch=B, max_num=args.max_num, device_id=local_rank, precision=precision))
            run(0, PREROLL + 1)
            barrier()
            t2 = time.perf_counter()
            ns = max(nb, min(args.steps, 40))
            run(PREROLL + 1, ns)
            barrier()
            dt2 = time.perf_counter() - t2
            if dist is not None:
                tt = torch.tensor([dt2], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt2 = float(tt.item())
            ctx.close()
            ctx = main_ctx
            return {"value": round(B * ns * world / dt2, 1), "unit": "frames/s", "steps": ns, "dtype": dtype, "what": what}

        split = split_pass(3, "f16x3",
                           "the same step with ppg_config.precision = 3: the 3x3 convolutions (98 % of the FLOPs) on the fp16 "
                           "MFMA with every operand scaled by a power of two and split into fp16 hi + lo, three products, fp32 "
                           "accumulate (about 22 significant bits per operand); tensors within 4e-6 of the fp32 path, "
                           "descriptors within 1e-4, key points equal up to ties within 4e-5 in score; integer outputs are "
                           "not bit-exact against the CPU restatement, which only the fp32 mode is")
        split_bf16 = split_pass(1, "bf16x3",
                                "the same step with ppg_config.precision = 1 (the 'bf16 conv / fp32 descriptors' configuration, in "
                                "its split form): the convolutions on the bf16 MFMA with operands split into bf16 hi + lo, three "
                                "products, fp32 accumulate; descriptors within 1e-3 of the fp32 path, key-point sets equal up "
                                "to near-ties; not bit-exact on the integer outputs")
        plain_bf16 = split_pass(2, "bf16",
                                "the same step with ppg_config.precision = 2 (the 'bf16 conv / fp32 descriptors' configuration, "
                                "taken literally): plain bf16 operands, one product, fp32 accumulate.  OUTSIDE the parity bar - "
                                "descriptor L2 error up to 5e-2 against the fp32 path, a few per cent of the key points differ - "
                                "and reported for that deviation only, never as the number of record")

    if rank == 0:
        dom_ms, dom_launches = prof["conv1a"]
        flops_per_launch = fl["conv1a"] / max(dom_launches, 1)
        avg_launch_s = dom_ms * 1e-3 / max(dom_launches, 1)
        achieved = flops_per_launch / avg_launch_s / 1e12 if dom_launches else 0.0
        o_ms, o_launches = prof_all["conv3x3"]
        other_tf = fl_all["conv3x3"] / (o_ms * 1e-3) / 1e12 if o_launches else None
        all_conv_tf = (fl_all["conv3x3"] + fl_all["conv1a"] + fl_all["conv_other"]) / \
            ((o_ms + prof_all["conv1a"][0] + prof_all["conv_other"][0]) * 1e-3) / 1e12
        frames_total = B * args.steps * world
        # roofline of the dominant kernel in algorithmic FLOPs: the bf16 split executes three bf16 MFMAs per product
        peak = (FP32_MFMA_PEAK_TFLOPS, BF16_MFMA_PEAK_TFLOPS / 3.0, BF16_MFMA_PEAK_TFLOPS, BF16_MFMA_PEAK_TFLOPS / 3.0)[args.precision]
        cnn_txt = ("CNN fp32 MFMA", "CNN 3x3 layers on the bf16 MFMA with hi/lo split operands (3 products), fp32 accumulate",
                   "CNN 3x3 layers on the bf16 MFMA (plain bf16 operands), fp32 accumulate",
                   "CNN 3x3 layers on the fp16 MFMA with scaled hi/lo split operands (3 products), fp32 accumulate")[args.precision]
        kname = ("conv_mfma_kernel<3,2,8,true,32>", "conv_bf16_kernel<2,true,32,3>", "conv_bf16_kernel<2,true,32,1>",
                 "conv_bf16_kernel<2,true,32,3,true>")[args.precision]
        # the large 3x3 layers of mode 1 run wave-specialised by default (PPG_CONV_WS=1: modes 2 and 3 too, =0: none)
        ws_env = os.environ.get("PPG_CONV_WS")
        if args.precision and (ws_env == "1" or (ws_env is None and args.precision == 1)):
            kname = ("", "conv_ws_kernel<true,32,3,false>", "conv_ws_kernel<true,32,1,false>", "conv_ws_kernel<true,32,3,true>")[args.precision]
        out = {
            "metric": f"front-end frames/sec (extract+describe+match) @{W}x{H}",
            "value": round(frames_total / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_steps": preroll,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("f32", "bf16x3", "bf16", "f16x3")[args.precision], "data": "synthetic",
            "config": {"workload": f"{W}x{H} synthetic '{args.kind}' scenes, {B} camera streams per GPU per step, "
                                   f"full extract ({cnn_txt} + NMS + line graph + descriptors, JUNCTION_MAX_NUM={args.max_num}) "
                                   f"+ all-pairs SearchForInitialization match vs previous frame; " + ("input handed over as host buffers" if args.host_input else "input resident in HBM"),
                       "batch_per_gpu": B, "parallelism": f"stream-sharded x{world}, RCCL all-reduce(MIN) barrier per step",
                       "avg_keypoints": round(timed["kp"] / (B * args.steps), 1),
                       "avg_edges": round(timed["edges"] / (B * args.steps), 1),
                       "avg_matches": round(timed["matches"] / (B * args.steps), 1)},
            "roofline": {"bound": "mfma",
                         "kernel": kname + ": conv1a (1->64, on the fly) + conv1b (64->64, 3x3) + ReLU + "
                                   "2x2 max-pool at full resolution; 40 % of the front-end's FLOPs, the longest launch of a step",
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": int(dom_launches),
                         "flops_per_launch": flops_per_launch, "traffic": None,
                         "other_3x3_convolutions": {"kernel": ("conv_mfma_kernel<3,2,8,false,32|16>" if args.precision == 0 else "conv_bf16_kernel / conv_ws_kernel, not fused") +
                                                              " (eight launches per step: conv2a-conv4b, the heads' first convolutions in one, EdgeHeatmap block 1)",
                                                    "achieved": round(other_tf, 2) if other_tf else None,
                                                    "ms_per_step": round(o_ms / nb, 4)},
                         "all_convolutions_achieved": round(all_conv_tf, 2)},
            "device_ms_per_step": {k: round(v[0] / nb, 4) for k, v in prof_all.items()},
            "extract_only_frames_per_s": round(extract_only_fps, 1),
            "with_upload": with_upload,
            "batch_sweep": sweep,
            "matchers_ms": matchers,
            "f16_split_mode": split,
            "bf16_split_mode": split_bf16,
            "plain_bf16_mode": plain_bf16,
            "shutter": shutter_kind,
            "value_input": "images resident in HBM when the timed region starts; `with_upload` is the same step with every "
                           "image handed over in host memory (PCIe-inclusive)",
            "device_ms_note": f"from a second pass of {nb} steps with events around every kernel class (not the timed region)",
        }
        # HBM bytes per launch of the dominant kernel come from hardware counters, which only a profiler run can read:
        # profiles/traffic.json holds them per arithmetic mode (rocprofv3 --pmc passes of this very command,
        # tools/collect_profiles.sh); the line says where the figure is from.
        tr = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr):
            try:
                tj = json.load(open(tr))
                ent = tj.get("by_precision", {}).get(str(args.precision))
                if ent and tj.get("batch") == B and tj.get("width") == W and tj.get("height") == H:
                    out["roofline"]["traffic"] = ent.get("hbm_bytes_per_launch")
                    out["roofline"]["traffic_source"] = "profiles/traffic.json: " + tj.get("method", "rocprofv3 --pmc passes") + \
                        " - read from that file, not measured by this run"
                    out["roofline"]["algorithmic_bytes_per_launch"] = ent.get("algorithmic_bytes_per_launch")
            except Exception:
                pass
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(H, W, args.kind, args.max_num)
        print(json.dumps(out), flush=True)
    if hasattr(shutter, "close"):
        shutter.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
