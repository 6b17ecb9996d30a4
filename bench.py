#!/usr/bin/env python
"""bench.py -- frames/sec end-to-end detect -> track -> landmarks -> embed -> cluster on synthetic 1080p@25fps video
(BASELINE.json metric; workload = configs[1]: 1000 frames, 4 shots, ~8 faces/frame per GPU).

One "step" = one full pass of the hot path over the rank's 1000 frames already resident in HBM: HOG detection of every
frame, forward+backward correlation tracking per shot, landmarks + 128-D embedding of every tracked face, then (after an
all-gather of the embeddings when N > 1) one global clustering.  Weak scaling: every rank owns a 1000-frame range (cut at
shot boundaries) of one N x 1000-frame video.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = HOG filter scoring, HIP-event timed on the library's
stream inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample of the same frames, N = 1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))

FP32_PEAK_TFLOPS = 157.3   # dense fp32 (vector == f32 MFMA) peak, MI355X_MICROARCH.md chip table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--faces", type=int, default=8)
    ap.add_argument("--shots", type=int, default=4)
    ap.add_argument("--detect-batch", type=int, default=128, help="frames whose pyramids, features and scores are resident together (one scoring launch per batch)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): every rank owns --frames frames of an N x --frames video; strong: ONE video of --frames frames "
                         "(BASELINE.json configs[2]'s shape: a fixed video cut into N frame ranges at shot boundaries)")
    ap.add_argument("--cpu-frames", type=int, default=32, help="frames of the all-core CPU-oracle sample, centred on the first shot cut (0 = skip)")
    ap.add_argument("--cpu-frames-1t", type=int, default=4, help="frames of the single-thread CPU-oracle sample (same centre)")
    ap.add_argument("--no-host-ingest", action="store_true", help="skip the extra pass whose frames start in pinned host memory")
    ap.add_argument("--no-overlap", action="store_true", help="no GPU-feeding thread: every stage runs in the caller's thread, shot after shot")
    ap.add_argument("--small-models", action="store_true", help="debug only: reduced landmark model")
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("PVF_DIST_BACKEND", "nccl"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from pyannote_video_amd import synth, models, pipeline, dist as pdist
    from pyannote_video_amd.runtime import Context

    model_dir = os.path.join(tempfile.gettempdir(), "pvface_models_rank%d" % rank)
    lp, ep = models.ensure_synthetic_models(model_dir, small=args.small_models)

    if args.scaling == "strong" and world > 1:
        # ONE video of --frames frames; the ranks take contiguous shot ranges (dist.shard_shots), each rendering only its own frames
        whole = synth.SyntheticVideo(width=args.width, height=args.height, n_frames=args.frames, n_shots=max(args.shots, world),
                                     faces=args.faces, seed=20260925)
        all_times = [whole.timestamp(i) for i in range(whole.n_frames)]
        ranges = pipeline.split_into_shots(all_times, whole.shots())
        s0, s1 = pdist.shard_shots(ranges, world)[rank]
        i0, i1 = ranges[s0][0], ranges[s1 - 1][1]
        video = whole
        t_gen = time.time()
        frames_t = whole.frames_torch(device, indices=range(i0, i1))
        torch.cuda.synchronize()
        t_gen = time.time() - t_gen
        times = all_times[i0:i1]
        shots = whole.shots()[s0:s1]
        n_local = i1 - i0
    else:
        # this rank's frame range of the long video: its own faces/backgrounds (seed), timestamps continue across ranks
        video = synth.SyntheticVideo(width=args.width, height=args.height, n_frames=args.frames, n_shots=args.shots,
                                     faces=args.faces, seed=20260925 + rank)
        t_gen = time.time()
        frames_t = video.frames_torch(device)
        torch.cuda.synchronize()
        t_gen = time.time() - t_gen
        t_off = rank * args.frames / video.frame_rate
        times = [t_off + video.timestamp(i) for i in range(args.frames)]
        shots = [(t_off + a, t_off + b) for a, b in video.shots()]
        n_local = args.frames

    ctx = Context(device=local_rank)
    frames = [ctx.wrap_torch(frames_t[i]) for i in range(n_local)]
    pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=args.detect_batch, overlap=not args.no_overlap)

    def step():
        tm = {}
        t_step = time.perf_counter()
        res = pipe.run(frames, times, video.frame_rate, shots, timings=tm, cluster=False, last_shard=(rank == world - 1), reorder=(world == 1))
        T, ids, X, offsets = pdist.gather_rows(res["face_T"], res["face_id"], res["X"], len(res["tracks"]), device=device,
                                               file_T=res["file_T"] if world > 1 else None, file_id=res["file_id"] if world > 1 else None)
        t0 = time.perf_counter()
        labels = pdist.global_cluster(pipe.clustering, T, ids, X)
        tm["cluster_s"] = time.perf_counter() - t0
        tm["step_wall_s"] = time.perf_counter() - t_step
        return res, labels, tm

    ctxs = [ctx]

    def barrier():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    for c in ctxs:
        c.prof_reset()
        c.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    last = None
    prof_path = os.environ.get("PVF_PYPROF")
    if prof_path:
        import cProfile
        pr = cProfile.Profile()
        pr.enable()
    for _ in range(args.steps):
        last = step()
    if prof_path:
        pr.disable()
        import pstats
        with open(prof_path, "w") as f:
            pstats.Stats(pr, stream=f).sort_stats("cumulative").print_stats(45)
    barrier()
    elapsed = time.perf_counter() - t0
    for c in ctxs:
        c.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    res, labels, tm = last
    if os.environ.get("PVF_DUMP") and rank == 0:
        ident_of_track = {}
        np.savez_compressed(os.environ["PVF_DUMP"], X=res["X"], emb=res["embeddings"], face_id=res["face_id"], face_T=res["face_T"],
                            boxes=np.array(res["face_boxes"]), labels=np.array(sorted(labels.items())),
                            gt=np.array([[k, f, tr["ident"]] for k, shot in enumerate(video.tracks) for f, tr in enumerate(shot)]),
                            track_first=np.array([[i, tr[0][0]] + list(tr[0][1]) for i, tr in enumerate(res["tracks"])]))
    total_frames = (args.frames if (args.scaling == "strong" and world > 1) else args.frames * world) * args.steps
    fps = total_frames / elapsed

    fam = {}
    for name in ("pyramid", "fhog", "score", "chip", "ert", "conv", "dsst", "pdist", "hac"):
        ms, n = 0.0, 0
        for c in ctxs:
            a, b = c.prof_get(name)
            ms += a; n += b
        fam[name] = {"ms": round(ms, 3), "launches": int(n)}

    if rank != 0:
        return
    geo = pipeline.detector_geometry(args.height, args.width)
    positions = sum(g[4] for g in geo)
    flop_per_frame = positions * 3100 * 5 * 2.0          # 10x10 cells x 31 planes, 5 filters, FMA = 2 flop
    n_score_frames = n_local * args.steps
    score_ms = fam["score"]["ms"]
    launches = max(fam["score"]["launches"], 1)
    achieved = (flop_per_frame * n_score_frames / (score_ms * 1e-3)) / 1e12 if score_ms > 0 else 0.0
    roofline = {"kernel": "score_mfma_rows_ml_k<4> (HOG filter scoring of every pyramid level of a %d-frame batch, 5 filters x 3100 MAC per position)" % args.detect_batch, "bound": "mfma",
                "achieved": round(achieved, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / FP32_PEAK_TFLOPS, 4), "traffic": None,
                "avg_launch_ms": round(score_ms / launches, 4),
                "flop_per_launch": flop_per_frame * n_score_frames / launches}

    # HBM traffic of the dominant kernel comes from a separate rocprofv3 --pmc pass (counters cannot be read inside this process);
    # the committed measurement is attached when it was taken on this configuration.
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_score.json")))
        if pm["detect_batch"] == args.detect_batch and pm["frame"] == "%dx%d" % (args.width, args.height):
            roofline["traffic"] = pm["traffic_bytes_per_launch"]
            roofline["traffic_source"] = pm["source"]
            roofline["algorithmic_bytes_per_launch"] = sum(g[2] * g[3] for g in geo) * 128.0 * n_score_frames / launches   # features read once
    except Exception:
        pass

    cpu, parity = None, None
    if world == 1 and args.cpu_frames > 0:
        cpu, parity = cpu_baseline_and_parity(video, frames_t, ctx, pipe, lp, ep, args)
    host = None
    if world == 1 and not args.no_host_ingest:
        host = host_ingest_pass(ctx, pipe, frames_t, times, video, shots, args)

    n_clusters = len(set(labels.values()))
    out = {
        "metric": "frames/sec end-to-end detect->embed->cluster, 1080p@25fps",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
        "vs_baseline": None, "dtype": "f32 (detector, embedder) / f64 (tracker, clustering) / u8 frames",
        "data": "synthetic (procedural faces on low-pass backgrounds, seeded; synthetic model weights of dlib's shapes)",
        "config": {"workload": "configs[1]: synthetic %dx%d 25 fps, %d frames, %d shots, %d faces/frame %s, frames resident in HBM"
                               % (args.width, args.height, args.frames, args.shots, args.faces,
                                  "in total, one video cut into shot ranges" if (args.scaling == "strong" and world > 1) else "per GPU"),
                   "detect_every": 0, "upsample": 1, "tracking": "forward+backward DSST, CLI defaults (overlap 0.5, conf 10, gap 1.0)",
                   "parallelism": "shot-range sharding x%d + all-gather of track embeddings" % world if world > 1 else "single GPU",
                   "detect_batch": args.detect_batch},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity": parity,
        "host_ingest": host,
        "stage_seconds_last_step": {k: round(v, 3) for k, v in tm.items()},
        "kernel_families_ms": fam,
        "results": {"tracks": len(res["tracks"]), "faces_embedded": int(len(res["face_T"])), "clusters": n_clusters,
                    "identities_in_video": len(set(tr["ident"] for shot in video.tracks for tr in shot))},
        "setup_seconds": {"generate_frames_in_hbm": round(t_gen, 1)},
    }
    print(json.dumps(out))


def host_ingest_pass(ctx, pipe, frames_t, times, video, shots, args):
    """Outside the timed region (`value` is quoted with the frames resident in HBM): the same step with the frames starting in pinned
    host memory -- where a decoder would leave them -- and reaching HBM through the ingest ring: one asynchronous copy per frame on the
    copy stream, kernels waiting per frame on the device, so the uploads of later shots run beside the detector of the first ones."""
    import numpy as np
    n = len(times)
    h, w = int(frames_t.shape[1]), int(frames_t.shape[2])
    ring = ctx.ingest_ring(h, w, depth=n)                  # the whole clip "decoded" into pinned slots, once
    for i in range(n):
        np.copyto(ring.slot(), frames_t[i].cpu().numpy())
    bytes_total = float(n) * h * w * 3

    def submit_all():
        out = []
        for i in range(n):
            ring.slot()                                    # slot i again (its bytes are still there); waits for its previous upload
            out.append(ring.submit())
        return out
    ctx.sync()
    t0 = time.perf_counter()
    dev = submit_all()
    ring.wait()
    t_copy = time.perf_counter() - t0                      # uploads alone: what PCIe delivers
    for f in dev:
        f.release()
    best = None
    for _ in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        dev = submit_all()
        res = pipe.run(dev, times, video.frame_rate, shots, cluster=True)
        ctx.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        for f in dev:
            f.release()
    ring.close()
    return {"value": round(n / best, 2), "unit": "frames/s", "frames_start_in": "pinned host memory (ingest ring, async H2D on a copy stream, per-frame waits on the device)",
            "pcie_GBps_uploads_alone": round(bytes_total / t_copy / 1e9, 2), "ms_per_step": round(1000 * best, 2), "tracks": len(res["tracks"])}


def cpu_baseline_and_parity(video, frames_t, ctx, pipe, lp, ep, args):
    """Outside the timed region: the CPU oracle (a port: dlib itself is not installable here) on a bounded sample of the SAME video
    -- a window of frames centred on the first shot cut, so that it holds a shot boundary -- timed with all host cores and with one
    thread, and the product pipeline on the same window compared with it: the parity gate of this run."""
    import concurrent.futures
    import numpy as np
    from pyannote_video_amd import models, pipeline
    from oracle import oracle, ref_flow
    nproc = oracle.usable_cpus(cap=1024)          # CPUs this process owns (affinity mask and cgroup quota), not the host's processor count
    cut = video.shot_bounds[1] if video.n_shots > 1 else video.n_frames // 2
    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    sp = oracle.ShapePredictor(models.load_model_file(lp, "shape_predictor"))
    emb = oracle.Embedder(models.load_model_file(ep, "embedder"))
    tabs = models.dsst_tables()

    def window(n):
        i0 = max(0, min(cut - n // 2, video.n_frames - n))
        idx = list(range(i0, i0 + n))
        shots = [(a, b) for a, b in video.shots() if b > video.timestamp(idx[0]) and a <= video.timestamp(idx[-1])]
        return idx, [video.timestamp(i) for i in idx], shots

    def note(msg):
        if os.environ.get("PVF_VERBOSE"):
            sys.stderr.write("[bench %.1fs] %s\n" % (time.perf_counter() - t_begin, msg))
            sys.stderr.flush()
    t_begin = time.perf_counter()

    def oracle_flow(idx, times, shots, threads):
        oracle.lib().pvo_set_threads(threads)
        frames = [np.ascontiguousarray(frames_t[i].cpu().numpy()) for i in idx]      # the bytes the GPU path sees
        pool = concurrent.futures.ThreadPoolExecutor(min(threads, 32)) if threads > 1 else None
        note("oracle flow on %d frames, %d threads" % (len(idx), threads))
        t0 = time.perf_counter()
        tracks = ref_flow.track_video(frames, times, shots, det, lambda: oracle.Tracker(tabs), video.frame_rate,
                                      min_conf=pipeline.CLI_MIN_CONFIDENCE, ratio=pipeline.CLI_MIN_OVERLAP_RATIO, max_gap=pipeline.CLI_MAX_GAP, pool=pool)
        t_track = time.perf_counter() - t0
        note("  detect + tracking done (%.1f s)" % t_track)
        lm, em = ref_flow.extract(ref_flow.track_text(tracks), frames, times, sp, emb, pool=pool)
        labels = ref_flow.cluster(em, 0.6)
        dt = time.perf_counter() - t0
        note("  whole flow done (%.1f s)" % dt)
        if pool is not None:
            pool.shutdown()
        return tracks, lm, em, labels, dt, t_track

    idx, times, shots = window(min(args.cpu_frames, video.n_frames))
    tracks, lm, em, labels, dt, t_track = oracle_flow(idx, times, shots, nproc)
    cpu = {"value": round(len(idx) / dt, 4), "unit": "frames/s", "cores": int(nproc), "kind": "port",
           "sample": "frames %d..%d of the same 1080p video (a window around the shot cut at frame %d), whole flow: detect + fwd/bwd tracking %.1f s "
                     "of %.1f s, then landmarks, embedding, clustering; OpenMP over pyramid rows, FHOG cell rows and scoring rows, trackers and "
                     "faces of a frame in a thread pool" % (idx[0], idx[-1], cut, t_track, dt)}
    if args.cpu_frames_1t > 0:
        idx1, times1, shots1 = window(min(args.cpu_frames_1t, video.n_frames))
        _, _, _, _, dt1, _ = oracle_flow(idx1, times1, shots1, 1)
        cpu["single_thread"] = {"value": round(len(idx1) / dt1, 4), "unit": "frames/s", "cores": 1,
                                "sample": "frames %d..%d, same flow, one thread (how the reference runs: one Python thread, single-threaded dlib)" % (idx1[0], idx1[-1])}
    # ---- parity gate: the product on the same window (frames already in HBM) vs the oracle flow above
    note("product pipeline on the parity window")
    dev_frames = [ctx.wrap_torch(frames_t[i]) for i in idx]
    res = pipe.run(dev_frames, times, video.frame_rate, shots)
    ref_e = np.array([[float(x) for x in line.split()[2:]] for line in em]).reshape(-1, 128)
    ref_pts = np.array([[float(x) for x in line.split()[2:]] for line in lm]).reshape(-1, 68, 2)
    w, h = video.frame_size
    ref_int = np.rint(ref_pts * np.array([w, h], np.float64)).astype(np.int64)      # 5 decimals of x / width resolve the integer point
    same_rows = len(ref_e) == len(res["embeddings"]) and [int(l.split()[1]) for l in em] == res["face_id"].tolist()
    parity = {
        "sample": "product pipeline vs CPU oracle flow on frames %d..%d (1080p, full landmark model, detect batch %d)" % (idx[0], idx[-1], args.detect_batch),
        "boxes": "exact" if res["tracks"] == tracks else "MISMATCH",           # every track row: time, detector / tracker box, status string
        "track_ids": "exact" if [len(t) for t in res["tracks"]] == [len(t) for t in tracks] and same_rows else "MISMATCH",
        "landmarks": "exact" if same_rows and np.array_equal(res["landmarks"].astype(np.int64), ref_int) else "MISMATCH",
        "embed_l2_max": float(np.linalg.norm(ref_e - res["embeddings"].astype(np.float64), axis=1).max()) if same_rows and len(ref_e) else None,
        "labels": "exact" if res["labels"] == labels else "MISMATCH",
        "tracks": len(tracks), "faces": int(len(ref_e)),
    }
    return cpu, parity


if __name__ == "__main__":
    main()
