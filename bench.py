#!/usr/bin/env python
"""bench.py -- frames/sec end-to-end detect -> track -> landmarks -> embed -> cluster on synthetic 1080p@25fps video
(BASELINE.json metric; workload = configs[1]: 1000 frames, 4 shots, ~8 faces/frame per GPU).

One "step" = one full pass of the hot path over the rank's 1000 frames already resident in HBM: HOG detection of every
frame, forward+backward correlation tracking per shot, landmarks + 128-D embedding of every tracked face, then (after an
all-gather of the embeddings when N > 1) one global clustering.  Weak scaling: every rank owns a 1000-frame range (cut at
shot boundaries) of one N x 1000-frame video.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = HOG filter scoring, HIP-event timed on the library's
stream inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample of the same frames, N = 1 only).

`--config` selects one of BASELINE.json's other configurations (the default, c2, is the one the metric is quoted on):
  c3  one long 1080p video streamed through the bounded-memory engine (configs[2]: --frames per rank, e.g. 22500 = 2 h / 8), frames
      delivered one by one from a resident 1000-frame clip played in a loop, released shot by shot; all-gather + one global clustering
  c4  64 independent 720p clips of 250 frames farmed over the ranks (configs[3]): per-clip clustering, no collective
  c5  4K, 50 fps, 40 faces per frame (configs[4]): the c2 step on 3840x2160 frames, with the peak HBM use of the run
Each prints its own JSON line (same contract, its own `metric` label).
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
F16_PEAK_TFLOPS = 2500.0    # dense f16 / bf16 MFMA peak, same table (the 5 PF headline figure includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0       # HBM3E, same table
PMC_KERNELS_FILE = "r06_pmc_kernels.json"  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the detector kernels, keyed on the hash of csrc/detect.hip + screen.hip
DTYPE_NOTE_DENSE = "f32 (detector, embedder) / f64 (tracker, clustering) / u8 frames"
DTYPE_NOTE = ("f32 (detector scores, embedder) / f64 (tracker, clustering) / u8 frames; an f16 screening pass with a proven error bound decides which windows the "
              "detector's exact f32 chain is evaluated for -- results identical to dense f32 scoring (`dense_scoring`)")
FAMILIES = ("pyramid", "fhog", "score", "score_screened", "chip", "ert", "conv", "dsst", "pdist", "hac")
EMBED_GFLOP_PER_FACE = 0.542               # the 29-conv ResNet on a 150 x 150 chip
TRACKER_GFLOP_PER_FRAME = 0.3              # DSST starts + updates of ~8 faces, both passes


def e2e_object(flop_score_per_frame, faces, frames, seconds, bytes_per_frame=None):
    """the whole step against BOTH roofs SURVEY.md section 8d names: arithmetic (scoring + embedding + tracker FLOPs of the frames processed)
    against the fp32 matrix peak -- the governing roof while the scoring runs dense in fp32 -- and the algorithmic HBM bytes of a frame
    against the HBM peak, the governing roof once the screening pass has moved the scoring sums to the f16 matrix cores"""
    g = (flop_score_per_frame * frames + EMBED_GFLOP_PER_FACE * 1e9 * faces + TRACKER_GFLOP_PER_FRAME * 1e9 * frames) / max(frames, 1) / 1e9
    tf = g * 1e9 * frames / seconds / 1e12 if seconds > 0 else 0.0
    o = {"gflop_per_frame": round(g, 3), "tflops": round(tf, 2), "dense_equivalent_frac_of_fp32_peak": round(tf / FP32_PEAK_TFLOPS, 4),
         "note": "scoring (positions x 3100 MAC x 5 filters) + %.3f GFLOP per embedded face + %.1f GFLOP/frame of tracker FFTs, over the timed steps' "
                 "wall time of the slowest rank (the scoring sums counted once per window, whichever matrix cores evaluate them); pyramid / FHOG / landmark work is byte-bound and not counted.  "
                 "DENSE-EQUIVALENT: the default path evaluates the scoring sums on the f16 pipe (screening) and the exact fp32 chain only for the listed windows, so this is not an achieved "
                 "fraction of the fp32 pipe -- the governing figure of the default path is frac_of_hbm_peak" % (EMBED_GFLOP_PER_FACE, TRACKER_GFLOP_PER_FRAME)}
    if bytes_per_frame:
        gbs = bytes_per_frame * frames / seconds / 1e9 if seconds > 0 else 0.0
        o.update({"hbm_bytes_per_frame": round(bytes_per_frame), "hbm_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                  "hbm_roof_frames_per_s": round(HBM_PEAK_GBS * 1e9 / bytes_per_frame, 1),
                  "hbm_note": "algorithmic bytes of one frame (DESIGN.md section 3): pyramid (frame read, every level written once and read once) + FHOG (levels read, 31 feature "
                              "planes written) + the feature maps read once by the scoring + tracker filters (2.06 MB per start / update) + chips and activations "
                              "of the embedder; against the 8 TB/s HBM peak (SURVEY.md section 8d prices this roof at 0.3 GB per frame and 6.3 TB/s achievable: about 21 k frames/s)"})
    return o


def frame_bytes(height, width, faces_per_frame):
    """algorithmic HBM bytes of one frame of the whole path (the byte roof of `e2e`)"""
    from pyannote_video_amd import pipeline
    geo = pipeline.detector_geometry(height, width)
    img = [g[0] * g[1] * 3.0 for g in geo]
    cells = [max(g[2] - 9, 0) * max(g[3] - 9, 0) for g in geo]
    pyramid = height * width * 3.0 + img[0] + sum(img[l - 1] + img[l] for l in range(1, len(img)))
    fhog = sum(img) + sum(cells) * 31 * 4.0
    score = sum(g[2] * g[3] for g in geo) * 128.0
    tracker = faces_per_frame * 2 * 2 * 2.06e6           # forward + backward pass: a start (filters written) and a first update (read) per detection
    embed = faces_per_frame * (150 * 150 * 3 + 6.5e6)    # chip + the activations of the 29 layers written once and read once or twice (NHWC f32, about 2.6 MB written per face; weights stay in cache)
    return pyramid + fhog + score + tracker + embed


def detector_hash():
    """sha256[:16] of the detector's sources WITHOUT their comments and blank space: the key of the attached counter measurements (a
    reworded comment is not a changed kernel)"""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in ("detect.hip", "screen.hip", "detect_ml.h", "fhog_dev.h"):
        src = open(os.path.join(ROOT, "pyannote-video_amd", "csrc", name), "r").read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        h.update(re.sub(r"\s+", " ", src).encode())
    return h.hexdigest()[:16]


def detector_rooflines(fam, height, width, frames_scored, detect_batch):
    """roofline objects of the detector's four kernels from the HIP-event family times of the timed steps; the first is the one that took
    the most time -- the step's dominant kernel (the other families are several kernels each, none of them as long).  Algorithmic work
    per frame from the level schedule:
      resize_rows_k   (20 launches per batch)  HBM: the frame read, every level written once and read once by the next stage
      fhog_split_ml_k (1 launch per batch)     HBM: every level image read once, 31 feature planes of every cell written once
      score_roll_k    (dense scoring)          fp32 MFMA: positions x 3100 MAC x 5 filters
      score_screen_k  (+ score_list_k)         f16 MFMA: the same sums (every window is scored), against the f16 peak"""
    from pyannote_video_amd import pipeline
    geo = pipeline.detector_geometry(height, width)
    img = [g[0] * g[1] * 3.0 for g in geo]
    cells = [max(g[2] - 9, 0) * max(g[3] - 9, 0) for g in geo]
    flop = sum(g[4] for g in geo) * 3100 * 5 * 2.0
    def per_launch(name):
        n = fam[name]["launches"]
        return int(round(frames_scored / float(n))) if n > 0 else detect_batch       # frames of one launch as launched (a 250-frame shot runs as 125 + 125, not 128 + 122)
    work = {"pyramid": ("resize_rows_k (every pyramid level of a %d-frame batch from the level above it, 20 launches)" % per_launch("pyramid"), "hbm",
                        height * width * 3.0 + img[0] + sum(img[l - 1] + img[l] for l in range(1, len(img)))),
            "fhog": ("fhog_split_ml_k (gradients, cell histograms and 31-plane features of every pyramid level of a %d-frame batch in one pass)" % per_launch("fhog"), "hbm",
                     sum(img) + sum(cells) * 31 * 4.0),
            "score": ("score_roll_k (HOG filter scoring of every pyramid level of a %d-frame batch, 5 filters x 3100 MAC per position)" % per_launch("score"), "mfma", flop),
            "score_screened": ("score_screen_k + score_list_k (every window of a %d-frame batch scored on the f16 matrix cores, the exact fp32 chain for the windows "
                               "within the error bound of the threshold)" % per_launch("score_screened"), "mfma", flop)}
    pm = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", PMC_KERNELS_FILE)))
        if not (pm["detect_batch"] == detect_batch and pm["frame"] == "%dx%d" % (width, height) and pm.get("detector_sha256_16") == detector_hash()):
            pm = None            # a changed kernel (or another configuration) drops the figure instead of carrying a stale one
    except Exception:
        pm = None
    out = []
    for name, (kernel, bound, per_frame) in work.items():
        ms, launches = fam[name]["ms"], fam[name]["launches"]
        if ms <= 0 or launches <= 0:
            continue
        if bound == "hbm":
            achieved, peak, unit = per_frame * frames_scored / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            peak = FP32_PEAK_TFLOPS if name == "score" else F16_PEAK_TFLOPS
            achieved, unit = per_frame * frames_scored / (ms * 1e-3) / 1e12, "TFLOP/s"
        o = {"kernel": kernel, "bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4), "traffic": None,
             "avg_launch_ms": round(ms / launches, 4), "launches": launches, "family_ms": ms,
             ("algorithmic_bytes_per_launch" if bound == "hbm" else "flop_per_launch"): per_frame * frames_scored / launches}
        k = pm and pm["kernels"].get(name)
        if k:
            o["traffic"] = k["traffic_bytes_per_launch"]
            o["traffic_attached_from_profiles_not_measured_in_this_run"] = pm["source"]
            if bound != "hbm":
                o["algorithmic_bytes_per_launch"] = sum(g[2] * g[3] for g in geo) * 128.0 * frames_scored / launches      # the feature maps read once
        out.append(o)
    out.sort(key=lambda o: -o["family_ms"])
    return out


def labels_digest(labels):
    """sha256[:16] of the sorted (track, label) pairs: two runs that print the same digest assigned every track to the same cluster"""
    import hashlib
    return hashlib.sha256(json.dumps(sorted((int(k), int(v)) for k, v in labels.items())).encode()).hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=("c2", "c3", "c4", "c5"), default="c2", help="BASELINE.json configuration (default c2 = configs[1], the one the metric is quoted on)")
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (c2: 1000, c3: 22500, c4: 250 per clip, c5: 500)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--fps", type=float, default=None)
    ap.add_argument("--faces", type=int, default=None)
    ap.add_argument("--shots", type=int, default=None)
    ap.add_argument("--clips", type=int, default=64, help="c4: number of clips in the farm")
    ap.add_argument("--detect-batch", type=int, default=None, help="frames whose pyramids, features and scores are resident together (one scoring launch per batch; 128 up to 1080p, 32 at 4K)")
    ap.add_argument("--detect-every", type=float, default=0.0, help="`--every` of the track verb: run the detector every that many seconds only (reference "
                    "tracking.py:383-386,425; 0 = every frame, the benched configuration); the trackers carry the faces in between")
    ap.add_argument("--dense-scoring", action="store_true", help="detector without its f16 screening pass: the exact fp32 chain for every window (csrc/screen.hip; same candidates, bit for bit)")
    ap.add_argument("--no-dense-leg", action="store_true", help="skip the three extra steps with the screening pass off (`dense_scoring` in the line)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the extra passes through the pyannote-face verbs (track / extract / cluster / process)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): every rank owns --frames frames of an N x --frames video; strong: ONE video of --frames frames "
                         "(BASELINE.json configs[2]'s shape: a fixed video cut into N frame ranges at shot boundaries)")
    ap.add_argument("--cpu-frames", type=int, default=100, help="frames of the all-core CPU-oracle sample, centred on the first shot cut (0 = skip)")
    ap.add_argument("--cpu-frames-1t", type=int, default=16, help="frames of the single-thread CPU-oracle sample (same centre)")
    ap.add_argument("--no-host-ingest", action="store_true", help="skip the extra pass whose frames start in pinned host memory")
    ap.add_argument("--preflight", action="store_true", help="--gpus N: before any frame is rendered every rank brings the communicator up, exchanges counts and "
                    "gathers 1 MB in uneven shares with every byte checked, each step under a 30 s watchdog and reported on stderr (dist.preflight)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE.json configs[2], [3], [4] the default line carries as `other_configs`")
    ap.add_argument("--other-configs-budget", type=float, default=240.0, help="seconds the `other_configs` runs may take together (each is a process of its own)")
    ap.add_argument("--no-overlap", action="store_true", help="no GPU-feeding thread: every stage runs in the caller's thread, shot after shot")
    ap.add_argument("--small-models", action="store_true", help="debug only: reduced landmark model")
    ap.add_argument("--parity-seed", type=int, default=None, help="seed of the extra 8-frame parity window placed at random in the clip (default: from the clock; printed in the line)")
    ap.add_argument("--distinct-clips", type=int, default=6, help="c3: how many differently seeded 1000-frame clips (identities drawn from a pool of 250) the long video "
                    "cycles through; 23 makes every loop of the default 22 500 frames its own clip (143 GB resident source)")
    ap.add_argument("--cluster-check-frames", type=int, default=-1, help="c3: the tracks of the first that many frames are clustered again by the CPU oracle "
                    "(default -1: ALL tracks of the range -- 720 tracks, 180 000 rows at the default 22 500 frames: about a minute of the host's cores; 0: skip)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST SWITCH: with --gpus N on a box with fewer than N devices, run the N ranks anyway, all on device 0 (gloo rendezvous, "
                         "torch.distributed collectives: two ranks of one RCCL communicator cannot share a device).  Exercises the launcher, the "
                         "shot-range sharding, the gather and the global clustering; the line says `oversubscribed`: its value is NOT a scaling figure")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the ranks this process launches (0: a free one)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)             # this process becomes the launcher: N ranks of this script, one per GPU
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d "
                         "bench.py --gpus %d ...), or run `python bench.py --gpus %d` and let it launch them\n"
                         % (args.gpus, os.environ["WORLD_SIZE"], args.gpus, args.gpus, args.gpus))
        sys.exit(2)
    defaults = {"c2": dict(frames=1000, width=1920, height=1080, fps=25.0, faces=8, shots=4, detect_batch=128),
                "c3": dict(frames=22500, width=1920, height=1080, fps=25.0, faces=8, shots=4, detect_batch=128),
                "c4": dict(frames=250, width=1280, height=720, fps=25.0, faces=8, shots=2, detect_batch=128),
                "c5": dict(frames=500, width=3840, height=2160, fps=50.0, faces=40, shots=2, detect_batch=32)}[args.config]
    for k, v in defaults.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.config == "c5" and args.cpu_frames == 100:
        args.cpu_frames, args.cpu_frames_1t = 8, 0          # a 4K frame costs the CPU oracle four 1080p frames

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    if args.oversubscribe and world > n_dev:
        local_rank = local_rank % max(n_dev, 1)          # several ranks per device (test switch)
    elif local_rank >= n_dev:
        sys.stderr.write("bench.py: rank %d needs device %d but this box shows %d GPU(s); --gpus N wants N devices "
                         "(--oversubscribe is the 1-GPU test switch)\n" % (rank, local_rank, n_dev))
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("PVF_DIST_BACKEND", "nccl"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from pyannote_video_amd import synth, models, pipeline, dist as pdist
    from pyannote_video_amd.runtime import Context
    if args.preflight or world > 1:
        # the first thing a multi-GPU run does (N > 1 has never run on hardware the builder could see): if the exchange step cannot work
        # the job says so here, per rank, within seconds -- not after the frames were rendered and a step ran into a silent collective
        pdist.preflight()

    model_dir = os.path.join(tempfile.gettempdir(), "pvface_models_rank%d" % rank)
    lp, ep = models.ensure_synthetic_models(model_dir, small=args.small_models)
    if args.config == "c4":
        return bench_farm(args, rank, local_rank, world, device, lp, ep)
    if args.config == "c3":
        return bench_stream(args, rank, local_rank, world, device, lp, ep)

    if args.scaling == "strong" and world > 1:
        # ONE video of --frames frames; the ranks take contiguous shot ranges (dist.shard_shots), each rendering only its own frames
        whole = synth.SyntheticVideo(width=args.width, height=args.height, n_frames=args.frames, n_shots=max(args.shots, world),
                                     faces=args.faces, seed=20260925, frame_rate=args.fps)
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
                                     faces=args.faces, seed=20260925 + rank, frame_rate=args.fps)
        t_gen = time.time()
        frames_t = video.frames_torch(device)
        torch.cuda.synchronize()
        t_gen = time.time() - t_gen
        t_off = rank * args.frames / video.frame_rate
        times = [t_off + video.timestamp(i) for i in range(args.frames)]
        shots = [(t_off + a, t_off + b) for a, b in video.shots()]
        n_local = args.frames

    ctx = Context(device=local_rank)
    if args.dense_scoring:
        ctx.detector_screening(False)
    frames = [ctx.wrap_torch(frames_t[i]) for i in range(n_local)]
    pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=args.detect_batch, overlap=not args.no_overlap, detect_every=args.detect_every)
    # the step needs the float32 descriptors (gathered, clustered on the device), not the float64 host copy of the clustering's table
    pipe.return_table = bool(os.environ.get("PVF_DUMP"))

    def step():
        tm = {}
        t_step = time.perf_counter()
        res = pipe.run(frames, times, video.frame_rate, shots, timings=tm, cluster=False, last_shard=(rank == world - 1), reorder=(world == 1))
        t0 = time.perf_counter()
        T, ids, X, offsets = pdist.gather_rows(res["face_T"], res["face_id"], res["embeddings"], len(res["tracks"]), device=device,
                                               file_T=res["file_T"] if world > 1 else None, file_id=res["file_id"] if world > 1 else None)
        tm["exchange_s"] = time.perf_counter() - t0          # the all-gather of the 528-byte face rows (waits for the slowest rank's shard)
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
    hbm = HbmSampler(ctx)
    scr_before = ctx.detector_screening_stats() if not args.dense_scoring else None
    t0 = time.perf_counter()
    last = None
    prof_path = os.environ.get("PVF_PYPROF")
    if prof_path:
        import cProfile
        pr = cProfile.Profile()
        pr.enable()
    frames_scored_timed = 0
    for _ in range(args.steps):
        last = step()
        frames_scored_timed += int(pipe.last_engine.stats.get("frames_detected", 0))
    if prof_path:
        pr.disable()
        import pstats
        with open(prof_path, "w") as f:
            pstats.Stats(pr, stream=f).sort_stats("cumulative").print_stats(45)
    barrier()
    elapsed = time.perf_counter() - t0
    hbm.stop()
    scr_after = ctx.detector_screening_stats() if not args.dense_scoring else None
    timed_engine = pipe.last_engine
    for c in ctxs:
        c.prof_enable(False)
    elapsed_local = elapsed
    elapsed = max_over_ranks(elapsed, world, device)
    res, labels, tm = last
    per_rank = None
    if world > 1:
        # what a bad scaling curve is read from without a second run: every rank's own time, its share, and where its last step went
        import torch.distributed as tdist
        mine = {"rank": rank, "ms_per_step": round(1000.0 * elapsed_local / args.steps, 2), "frames": int(n_local), "tracks": len(res["tracks"]),
                "faces": int(len(res["face_T"])), "last_step_s": {k: round(float(tm.get(k, 0.0)), 4) for k in ("track_s", "extract_s", "exchange_s", "cluster_s", "step_wall_s")}}
        per_rank = [None] * world
        tdist.all_gather_object(per_rank, mine)
    if os.environ.get("PVF_DUMP") and rank == 0:
        ident_of_track = {}
        np.savez_compressed(os.environ["PVF_DUMP"], X=res["X"], emb=res["embeddings"], face_id=res["face_id"], face_T=res["face_T"],
                            boxes=np.array(res["face_boxes"]), labels=np.array(sorted(labels.items())),
                            gt=np.array([[k, f, tr["ident"]] for k, shot in enumerate(video.tracks) for f, tr in enumerate(shot)]),
                            track_first=np.array([[i, tr[0][0]] + list(tr[0][1]) for i, tr in enumerate(res["tracks"])]))
    total_frames = (args.frames if (args.scaling == "strong" and world > 1) else args.frames * world) * args.steps
    fps = total_frames / elapsed

    def families():
        fam = {}
        for name in FAMILIES:
            ms, n = 0.0, 0
            for c in ctxs:
                a, b = c.prof_get(name)
                ms += a; n += b
            fam[name] = {"ms": round(ms, 3), "launches": int(n)}
        return fam
    fam = families()

    if rank != 0:
        return
    geo = pipeline.detector_geometry(args.height, args.width)
    positions = sum(g[4] for g in geo)
    flop_per_frame = positions * 3100 * 5 * 2.0          # 10x10 cells x 31 planes, 5 filters, FMA = 2 flop
    # frames the detector actually scored inside the timed steps (with --detect-every only every k-th frame is: counting all frames
    # would print a `frac` above 1); the engine counts them, summed over the ranks' steps on rank 0 only (every rank does the same work)
    n_score_frames = frames_scored_timed if frames_scored_timed is not None else n_local * args.steps
    rl = detector_rooflines(fam, args.height, args.width, n_score_frames, args.detect_batch)
    roofline, roofline_other = rl[0], rl[1:]

    # the same steps with the screening pass off (the exact fp32 chain for every window): what the screening buys, that the results are
    # the same, and the dense kernel's own roofline -- two timed steps after one untimed
    dense = None
    if world == 1 and not args.dense_scoring and not args.no_dense_leg and args.detect_every == 0.0:
        ctx.detector_screening(False)
        step()
        ctx.prof_reset(); ctx.prof_enable(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(2):
            d_res, d_labels, _tm = step()
        barrier()
        d_elapsed = time.perf_counter() - t0
        ctx.prof_enable(False)
        ctx.detector_screening(True)
        d_rl = [o for o in detector_rooflines(families(), args.height, args.width, n_local * 2, args.detect_batch) if o["kernel"].startswith("score_roll_k")]
        dense = {"value": round(n_local * 2 / d_elapsed, 2), "unit": "frames/s", "steps": 2, "ms_per_step": round(1000.0 * d_elapsed / 2, 2),
                 "same_tracks_faces_and_labels_as_the_timed_steps": bool(labels_digest(d_labels) == labels_digest(labels) and len(d_res["tracks"]) == len(res["tracks"])
                                                                      and np.array_equal(np.asarray(d_res["face_boxes"]), np.asarray(res["face_boxes"]))
                                                                      and np.array_equal(np.asarray(d_res["face_T"]), np.asarray(res["face_T"]))),
                 "roofline": d_rl[0] if d_rl else None,
                 "note": "pvf_detector_screening(ctx, 0): score_roll_k evaluates the exact chain of 3100 fmaf for every window on the fp32 matrix cores; the timed steps "
                         "screen every window on the f16 matrix cores first and run that chain for the listed ones only (csrc/screen.hip) -- same candidates, bit for bit"}

    # the same clip four times through ONE engine run (FacePipeline.run_many, what configs[3] is timed on): a step above ends with its
    # pipeline draining -- the last shots' tracker and extraction work alone on the GPU, then the clustering, then the host's result
    # assembly before the next step's first kernel -- and a deployment that has the next video waiting does not pay that drain.  NOT
    # `value` (whose steps run one after the other, each complete before the next starts); outside the timed region.
    back_to_back = None
    if world == 1 and not args.dense_scoring and not args.no_dense_leg and args.detect_every == 0.0 and args.config == "c2":
        clips = [dict(frames=frames, times=times, frame_rate=video.frame_rate, shots=shots) for _ in range(4)]
        pipe.run_many(clips)
        barrier()
        t0 = time.perf_counter()
        many = pipe.run_many(clips)
        barrier()
        b_elapsed = time.perf_counter() - t0
        same = all(labels_digest(r["labels"]) == labels_digest(labels) and len(r["tracks"]) == len(res["tracks"])
                   and np.array_equal(np.asarray(r["face_boxes"]), np.asarray(res["face_boxes"]))
                   and np.array_equal(np.asarray(r["face_T"]), np.asarray(res["face_T"])) for r in many)
        back_to_back = {"value": round(len(clips) * n_local / b_elapsed, 2), "unit": "frames/s", "clips": len(clips), "ms_per_clip": round(1000.0 * b_elapsed / len(clips), 2),
                        "same_tracks_faces_and_labels_as_the_timed_steps": bool(same),
                        "note": "FacePipeline.run_many: four copies of the clip as four jobs of one engine run -- the detector of clip i + 1 runs beside the tracker, "
                                "extraction and clustering tail of clip i; every clip clustered on its own, as in a timed step"}

    # ---- the WHOLE clip against the CPU oracle: the timed steps' own result (tracks, faces, landmarks, descriptors, labels) and the
    # detector's raw candidates of every frame against the fixture the oracle flow wrote for all 1000 frames (tests/golden/c2_full.npz,
    # made by tests/golden/make_full_clip.py: minutes of CPU, so it is frozen, not recomputed here); the dense leg the same way
    full_clip = None
    if world == 1 and args.detect_every == 0.0 and not args.small_models:
        # configs[1]: the whole clip; configs[4]: the clip's first shot (250 of its 500 4K frames: the oracle needs an hour for them)
        full_clip = full_clip_parity(ctx, frames, res, labels, args, d_res=d_res if dense else None, d_labels=d_labels if dense else None,
                                     name="c2_full" if args.config == "c2" else "c5_shot0")

    cpu, parity = None, None
    if world == 1 and args.cpu_frames > 0:
        cpu, parity = cpu_baseline_and_parity(video, frames_t, ctx, pipe, lp, ep, args)
    if full_clip is not None:
        parity = dict(parity or {}, full_clip=full_clip)
        if cpu is not None:
            cpu["whole_clip_fixture"] = {k: full_clip.get(k) for k in ("fixture", "frames", "all_exact", "raw_candidates", "embed_l2_max", "oracle_seconds", "oracle_threads")}
    host = None
    if world == 1 and not args.no_host_ingest and args.config == "c2":
        host = host_ingest_pass(ctx, pipe, frames_t, times, video, shots, args)
    dropin = None
    if world == 1 and not args.no_dropin and args.config == "c2" and args.detect_every == 0.0:
        dropin = dropin_cli_pass(ctx, frames, video, lp, ep, fps, res, labels)

    other = None
    if (world == 1 and args.config == "c2" and not args.no_other_configs and not args.dense_scoring and not args.small_models and args.detect_every == 0.0
            and args.frames == 1000 and args.cpu_frames > 0):
        other = other_configs_pass(args)

    n_clusters = len(set(labels.values()))
    label = "1080p@25fps" if args.config == "c2" else "%dx%d@%gfps (BASELINE.json configs[4])" % (args.width, args.height, args.fps)
    out = {
        "metric": "frames/sec end-to-end detect->embed->cluster, %s" % label,
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
        "vs_baseline": None, "dtype": DTYPE_NOTE if not args.dense_scoring else DTYPE_NOTE_DENSE,
        "data": "synthetic (procedural faces on low-pass backgrounds, seeded; synthetic model weights of dlib's shapes)",
        "config": {"workload": "configs[%d]: synthetic %dx%d %g fps, %d frames, %d shots, %d faces/frame %s, frames resident in HBM"
                               % (1 if args.config == "c2" else 4, args.width, args.height, args.fps, args.frames, args.shots, args.faces,
                                  "in total, one video cut into shot ranges" if (args.scaling == "strong" and world > 1) else "per GPU"),
                   "detect_every": args.detect_every, "upsample": 1, "tracking": "forward+backward DSST, CLI defaults (overlap 0.5, conf 10, gap 1.0)",
                   "parallelism": "shot-range sharding x%d + all-gather of track embeddings" % world if world > 1 else "single GPU",
                   "detect_batch": args.detect_batch, "collective": pdist.collective_name(),
                   "devices": min(world, n_dev), "oversubscribed": bool(world > n_dev)},
        "roofline": roofline,
        "roofline_other": roofline_other,
        "dense_scoring": dense,
        "back_to_back": back_to_back,
        "e2e": e2e_object(flop_per_frame if args.detect_every == 0.0 else flop_per_frame * n_score_frames / max(n_local * args.steps, 1),
                          int(len(res["face_T"])) * args.steps, n_local * args.steps, elapsed,
                          bytes_per_frame=frame_bytes(args.height, args.width, len(res["face_T"]) / float(max(n_local, 1))) if args.detect_every == 0.0 else None),
        "cpu_baseline": cpu,
        "parity": parity,
        "host_ingest": host,
        "dropin_cli": dropin,
        "other_configs": other,
        "hbm": hbm.report(frames_bytes=int(frames_t.numel()), engine=timed_engine),
        "stage_seconds_last_step": {k: round(v, 3) for k, v in tm.items()},
        "per_rank": per_rank,
        "per_rank_note": None if per_rank is None else "exchange_s = the all-gather of the face rows (includes waiting for the slowest rank's shard); cluster_s = this rank's "
                         "share of the pair distances (split by triangle area) + the all-gather of the distance rows + the agglomeration every rank repeats",
        "kernel_families_ms": fam,
        "screening": ctx.detector_screening_stats() if not args.dense_scoring else None,
        "kernel_families_note": "HIP-event time per family on the stream it runs on: the detector families (pyramid, fhog, score / score_screened) on one stream, the rest on the "
                                "other; the two streams run side by side, so the sum may exceed the steps' wall time, and a family's time includes "
                                "what it lost to the other stream's kernels",
        "results": {"tracks": len(res["tracks"]), "faces_embedded": int(len(res["face_T"])), "clusters": n_clusters,
                    "tracks_clustered_globally": len(labels), "labels_sha256_16": labels_digest(labels),
                    "identities_in_video": len(set(tr["ident"] for shot in video.tracks for tr in shot))},
        "setup_seconds": {"generate_frames_in_hbm": round(t_gen, 1)},
    }
    # (window, filter) pairs per frame that went through the exact fp32 chain inside the timed steps: what the screening pass's gain depends on
    out["listed_per_frame"] = (round((scr_after["listed"] - scr_before["listed"]) / float(frames_scored_timed), 2)
                               if scr_after and frames_scored_timed else None)
    out["dense_scoring_value"] = dense["value"] if dense else None
    if out["roofline"] is not None:
        out["roofline"]["step_with_dense_scoring_frames_per_s"] = out["dense_scoring_value"]
        out["roofline"]["screening_listed_pairs_per_frame"] = out.get("listed_per_frame")
    # last key of the line (the driver keeps the line's tail verbatim): the whole-clip verdict in one short string
    out["full_clip_parity"] = full_clip_summary(full_clip)
    if other is not None:
        out["full_clip_parity"] += " || other configs: " + "; ".join(("%s %s frames/s, %s" % (k, v.get("frames_per_s"), v.get("verdict"))) if v.get("frames_per_s") else ("%s: %s" % (k, v.get("verdict")))
                                                                        for k, v in other.items() if isinstance(v, dict))
    print(json.dumps(out))


def other_configs_pass(args):
    """BASELINE.json configs[2], [3], [4] at a size that fits a default run, each as a process of its own (`python bench.py --config ...`,
    this file), each ending with the whole-clip comparison against its committed CPU-oracle fixture (tests/golden/c3_clip0.npz,
    c4_clip0.npz, c5_shot0.npz):
      c5_shot0   the 4K 50 fps 40-face clip (500 frames, two steps); the fixture covers its first shot (250 frames) -- configs[4]
      c4_8clips  8 of the 64 independent 720p clips through one engine run                -- configs[3]
      c3_clip0   the long video's first 1000-frame clip, STREAMED (frames copied into fresh library buffers as a decoder would) -- configs[2]
    Not `value`; outside the timed region; the GPU is otherwise idle while they run (this process only waits)."""
    import subprocess
    runs = [("c5_shot0", ["--config", "c5", "--steps", "2", "--warmup", "1"]),
            ("c4_8clips", ["--config", "c4", "--clips", "8", "--steps", "2", "--warmup", "1"]),
            ("c3_clip0", ["--config", "c3", "--frames", "1000", "--distinct-clips", "1", "--steps", "1", "--warmup", "1", "--cluster-check-frames", "0"])]
    common = ["--gpus", "1", "--cpu-frames", "0", "--no-host-ingest", "--no-dropin", "--no-dense-leg", "--no-other-configs"]
    out = {}
    t_all = time.perf_counter()
    for name, extra in runs:
        left = args.other_configs_budget - (time.perf_counter() - t_all)
        if left < 20.0:
            out[name] = {"frames_per_s": None, "verdict": "skipped: the budget of %g s was used up" % args.other_configs_budget}
            continue
        t0 = time.perf_counter()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=left)
            line = [l for l in p.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                out[name] = {"frames_per_s": None, "verdict": "failed (rc %d): %s" % (p.returncode, p.stderr.decode("utf-8", "replace")[-300:])}
                continue
            d = json.loads(line[-1])
            fc = ((d.get("parity") or {}).get("full_clip") or {})
            out[name] = {"frames_per_s": d.get("value"), "ms_per_step": d.get("ms_per_step"), "steps": d.get("steps"), "workload": (d.get("config") or {}).get("workload"),
                         "fixture": fc.get("fixture"), "exact": fc.get("all_exact"), "verdict": d.get("full_clip_parity"),
                         "results": d.get("results"), "wall_s": round(time.perf_counter() - t0, 1)}
        except subprocess.TimeoutExpired:
            out[name] = {"frames_per_s": None, "verdict": "timed out after %.0f s" % left}
    # configs[4] names 10 000 tracks: the clustering stressor at that size (tools/c5_cluster.py) against the CPU oracle's frozen labels and
    # merge list (tests/golden/c5_cluster_T10000.npz)
    left = args.other_configs_budget - (time.perf_counter() - t_all)
    if left >= 20.0:
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "c5_cluster.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=left)
            rows = [json.loads(l) for l in p.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
            r = [x for x in rows if x.get("T") == 10000 and x.get("rows_per_track") == 10]
            if p.returncode == 0 and r and r[0].get("oracle_fixture"):
                fx = r[0]["oracle_fixture"]
                ok = fx["labels_equal_oracle"] and fx["merges_equal_oracle_in_order"] and fx["f32_path_labels_equal_oracle"]
                out["c5_cluster_T10000"] = {"frames_per_s": None, "pdist_ms": r[0]["pdist_ms"], "hac_ms": r[0]["hac_ms"], "N": r[0]["N"], "merges": r[0]["merges"],
                                            "oracle_fixture": fx, "fixture": "c5_cluster_T10000", "exact": bool(ok),
                                            "verdict": "clustering of 10 000 tracks (N = 1e5): labels and all %d merges %s the oracle fixture's" % (r[0]["merges"], "EQUAL" if ok else "DIFFER from")}
            else:
                out["c5_cluster_T10000"] = {"frames_per_s": None, "verdict": "failed (rc %d): %s" % (p.returncode, p.stderr.decode("utf-8", "replace")[-300:])}
        except subprocess.TimeoutExpired:
            out["c5_cluster_T10000"] = {"frames_per_s": None, "verdict": "timed out"}
    return out


def full_clip_parity(ctx, frames, res, labels, args, d_res=None, d_labels=None, name="c2_full", seed=20260925, identities=12, n_frames=None):
    """product (the timed steps' last result) against the whole-clip fixture of the CPU oracle flow; a note instead when the benched clip
    is not the fixture's (other --frames / size / faces) or the fixture is absent.  A fixture that covers the FIRST shots of the benched
    video (c3_clip0: the long video's first 1000 frames; c5_shot0: the 4K clip's first shot) is compared with that part of the result
    (golden.prefix_of: tracks, face rows, landmarks, descriptors, raw candidates; not the labels, which the longer run decides over more tracks)."""
    from oracle import golden
    import numpy as np
    mine = dict(width=args.width, height=args.height, n_frames=n_frames or args.frames, n_shots=args.shots, faces=args.faces, seed=seed, frame_rate=args.fps,
                identities=identities)
    if not golden.matches(name, **mine) or not golden.available(name):
        return {"fixture": None, "why": "no fixture for this clip (tests/golden/make_full_clip.py %s writes the one of the default configuration)" % name}
    g = golden.load(name)
    take = int(g["video"][2])
    prefix = take < len(frames) or name == "c3_clip0"
    out = golden.compare(g, res, labels, prefix=prefix)
    frames = frames[:take]
    batch = 125 if args.detect_batch >= 125 else args.detect_batch
    t0 = time.perf_counter()
    raw = ctx.detect_raw_many(frames, batch)
    out["raw_candidates"] = golden.compare_raw(g, [golden.raw_key(r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4]) for r in raw])
    out["raw_candidates_total"] = int(g["raw_counts"].sum())
    out["raw_pass_seconds"] = round(time.perf_counter() - t0, 3)
    if d_res is not None:
        dd = golden.compare(g, d_res, d_labels, prefix=prefix)
        out["dense_scoring_leg"] = "exact" if dd["all_exact"] else {k: dd[k] for k in ("tracks", "face_rows", "landmarks", "embed_l2_max", "labels")}
    if d_res is not None or name != "c2_full":
        ctx.detector_screening(False)
        rawd = ctx.detect_raw_many(frames, batch)
        ctx.detector_screening(True)
        out["raw_candidates_dense"] = golden.compare_raw(g, [golden.raw_key(r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4]) for r in rawd])
    out["oracle_seconds"] = round(float(g["oracle_seconds"]), 1)
    out["oracle_threads"] = int(g["oracle_threads"])
    out["all_exact"] = bool(out["all_exact"] and out["raw_candidates"] == "exact" and out.get("dense_scoring_leg", "exact") == "exact"
                            and out.get("raw_candidates_dense", "exact") == "exact")
    return out


def full_clip_summary(fc):
    if not fc or not fc.get("fixture"):
        return "no whole-clip fixture for this configuration"
    if fc["all_exact"]:
        return "%s EXACT over all %d frames%s: %d tracks, %d faces (rows, landmarks)%s, %d raw candidates%s; embed L2 max %.2e" % (
            fc["fixture"], fc["frames"], " (the run's first ones)" if fc.get("prefix_of_a_longer_run") else "", fc["n_tracks"], fc["n_faces"],
            "" if fc.get("prefix_of_a_longer_run") else ", labels", fc["raw_candidates_total"],
            " (screened and dense)" if "raw_candidates_dense" in fc else "", fc["embed_l2_max"])
    bad = [k for k in ("tracks", "face_rows", "landmarks", "labels", "raw_candidates", "dense_scoring_leg", "raw_candidates_dense") if fc.get(k, "exact") != "exact"]
    return "%s MISMATCH in %s (embed L2 max %s)" % (fc["fixture"], ", ".join(bad) or "embedding", fc.get("embed_l2_max"))


def max_over_ranks(seconds, world, device):
    """the slowest rank's time (the all-reduce runs where the job's backend lives: the device for nccl, host memory for gloo)"""
    if world <= 1:
        return seconds
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks (one process per GPU, torch.distributed.run on
    127.0.0.1) with this command line and hand their output through; rank 0 prints the JSON line.  Fewer than N visible devices is an
    error unless --oversubscribe (the 1-GPU test switch) was given."""
    import socket
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < args.gpus:
        if not args.oversubscribe:
            sys.stderr.write("bench.py: --gpus %d but this box shows %d GPU(s)\n" % (args.gpus, n_dev))
            sys.exit(2)
        env["PVF_DIST_BACKEND"] = "gloo"             # ranks that share a device cannot form an RCCL communicator
        env["PVF_DIST_COLLECTIVE"] = "torch"
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


class HbmSampler(object):
    """device memory in use (hipMemGetInfo through the library: every allocation on this GPU) sampled every 10 ms while the timed steps run"""

    def __init__(self, ctx, period=0.01):
        import threading
        self.ctx, self.period = ctx, period
        free, total = ctx.mem_info()
        self.total, self.before, self.min_free = total, total - free, free
        self._run = True
        self.th = threading.Thread(target=self._loop, name="hbm-sampler")
        self.th.start()

    def _loop(self):
        while self._run:
            free, _ = self.ctx.mem_info()
            self.min_free = min(self.min_free, free)
            time.sleep(self.period)

    def stop(self):
        self._run = False
        self.th.join()

    def report(self, frames_bytes=None, engine=None):
        peak = self.total - self.min_free
        out = {"peak_bytes_in_use": int(peak), "in_use_before_timed_region": int(self.before), "device_total_bytes": int(self.total),
               "note": "hipMemGetInfo sampled every 10 ms during the timed steps; includes the resident input frames, the models, the "
                       "batch's pyramids and feature maps, tracker filters, chips and activations"}
        if frames_bytes is not None:
            out["resident_input_frames_bytes"] = int(frames_bytes)
        if engine is not None:
            out["windowed_shots_last_step"] = int(engine.stats.get("windowed_shots", 0))
        return out


class ResidentVideo(object):
    """frames resident in HBM behind the iteration contract of the reference's Video (video.py:411-464): yields (t, DeviceFrame)"""

    def __init__(self, frames, frame_rate, size, t0_index=0):
        self.frames, self.frame_rate, self._size, self._frame_size, self.t0 = frames, float(frame_rate), size, size, int(t0_index)

    size = property(lambda self: self._size)

    @property
    def frame_size(self):
        return self._frame_size

    @frame_size.setter
    def frame_size(self, v):
        self._frame_size = tuple(int(x) for x in v)

    def __len__(self):
        return len(self.frames)

    def __iter__(self):
        for i, f in enumerate(self.frames):
            yield (self.t0 + i) / self.frame_rate, f


def dropin_cli_pass(ctx, frames, video, lp, ep, fps_pipeline, res, labels):
    """Outside the timed region: the same clip through the reference-named entry points -- the `track`, `extract` and `cluster` verbs one
    after the other (three passes over the frames, files in between, like scripts/pyannote-face.py:239-314), and the one-pass `process`
    verb -- with the frames resident in HBM as for `value`.  Their files must say what the timed step computed."""
    import numpy as np
    from pyannote_video_amd import cli, formats
    from pyannote_video_amd._core import Segment
    d = tempfile.mkdtemp(prefix="pvface_dropin_")
    n = len(frames)
    rv = ResidentVideo(frames, video.frame_rate, video.frame_size)
    shots = [Segment(a, b) for a, b in video.shots()]
    paths = {k: os.path.join(d, k + ".txt") for k in ("track", "landmarks", "embedding", "labels", "track1", "landmarks1", "embedding1", "labels1")}
    best = None
    for _ in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        cli.track(rv, shots, paths["track"], ctx=ctx)
        t1 = time.perf_counter()
        cli.extract(rv, lp, ep, paths["track"], paths["landmarks"], paths["embedding"], ctx=ctx)
        t2 = time.perf_counter()
        lab = cli.cluster(paths["embedding"], paths["labels"], ctx=ctx)
        t3 = time.perf_counter()
        cur = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
        best = cur if best is None or cur[0] < best[0] else best
    one = None
    for _ in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        r1 = cli.process(rv, shots, lp, ep, paths["track1"], paths["landmarks1"], paths["embedding1"], paths["labels1"], ctx=ctx)
        dt = time.perf_counter() - t0
        one = dt if one is None else min(one, dt)
    same = open(paths["track"]).read() == open(paths["track1"]).read() and open(paths["landmarks"]).read() == open(paths["landmarks1"]).read()
    track_same = [l for i, tr in enumerate(res["tracks"]) for l in formats.track_lines(i, tr)] == open(paths["track"]).readlines()
    return {"verbs_track_extract_cluster": {"value": round(n / best[0], 2), "unit": "frames/s", "seconds": {"track": round(best[1], 4), "extract": round(best[2], 4), "cluster": round(best[3], 4)},
                                            "passes_over_the_frames": 2},
            "verb_process_one_pass": {"value": round(n / one, 2), "unit": "frames/s"},
            "fraction_of_value": {"three_verbs": round(n / best[0] / fps_pipeline, 3), "process": round(n / one / fps_pipeline, 3)},
            "files": {"track_file_equals_timed_step": bool(track_same), "three_verbs_equal_one_pass": bool(same), "labels_equal_timed_step": lab == labels and r1["labels"] == labels},
            "frames_start_in": "HBM (resident, as for `value`); text files written with the reference's formats; best of 3 passes each"}


def _models_for_oracle(lp, ep):
    from pyannote_video_amd import models
    from oracle import oracle
    return (oracle.Detector(models.load_container(models.DEFAULT_DETECTOR)), oracle.ShapePredictor(models.load_model_file(lp, "shape_predictor")),
            oracle.Embedder(models.load_model_file(ep, "embedder")), models.dsst_tables())


def oracle_window_parity(frames_np, times, shots, frame_rate, size, res, lp, ep, threads=None, label=""):
    """the CPU oracle flow on a window of frames against the product's result on the same window: the parity dictionary + oracle seconds"""
    import concurrent.futures
    import numpy as np
    from pyannote_video_amd import pipeline
    from oracle import oracle, ref_flow
    det, sp, emb, tabs = _models_for_oracle(lp, ep)
    threads = threads or oracle.usable_cpus(cap=1024)
    oracle.lib().pvo_set_threads(threads)
    pool = concurrent.futures.ThreadPoolExecutor(min(threads, 32)) if threads > 1 else None
    t0 = time.perf_counter()
    tracks = ref_flow.track_video(frames_np, times, shots, det, lambda: oracle.Tracker(tabs), frame_rate,
                                  min_conf=pipeline.CLI_MIN_CONFIDENCE, ratio=pipeline.CLI_MIN_OVERLAP_RATIO, max_gap=pipeline.CLI_MAX_GAP, pool=pool)
    lm, em = ref_flow.extract(ref_flow.track_text(tracks), frames_np, times, sp, emb, pool=pool)
    labels = ref_flow.cluster(em, 0.6)
    dt = time.perf_counter() - t0
    if pool is not None:
        pool.shutdown()
    ref_e = np.array([[float(x) for x in line.split()[2:]] for line in em]).reshape(-1, 128)
    ref_pts = np.array([[float(x) for x in line.split()[2:]] for line in lm]).reshape(-1, 68, 2)
    w, h = size
    ref_int = np.rint(ref_pts * np.array([w, h], np.float64)).astype(np.int64)
    same_rows = len(ref_e) == len(res["embeddings"]) and [int(l.split()[1]) for l in em] == res["face_id"].tolist()
    parity = {"sample": label,
              "boxes": "exact" if res["tracks"] == tracks else "MISMATCH",
              "track_ids": "exact" if [len(t) for t in res["tracks"]] == [len(t) for t in tracks] and same_rows else "MISMATCH",
              "landmarks": "exact" if same_rows and np.array_equal(res["landmarks"].astype(np.int64), ref_int) else "MISMATCH",
              "embed_l2_max": float(np.linalg.norm(ref_e - res["embeddings"].astype(np.float64), axis=1).max()) if same_rows and len(ref_e) else None,
              "labels": "exact" if res["labels"] == labels else "MISMATCH",
              "tracks": len(tracks), "faces": int(len(ref_e))}
    return parity, dt, threads


def bench_farm(args, rank, local_rank, world, device, lp, ep):
    """BASELINE.json configs[3]: a batch of independent 720p clips farmed over the GPUs (round robin), every clip tracked, embedded and
    clustered on its own; no collective anywhere.  One step = this rank's clips through ONE engine run (FacePipeline.run_many)."""
    import numpy as np
    import torch
    from pyannote_video_amd import synth, pipeline, dist as pdist
    from pyannote_video_amd.runtime import Context
    ctx = Context(device=local_rank)
    if args.dense_scoring:
        ctx.detector_screening(False)
    mine = pdist.shard_clips(args.clips, world)[rank]
    t_gen = time.time()
    clips, videos, tensors = [], [], []
    for i in mine:
        v = synth.SyntheticVideo(width=args.width, height=args.height, n_frames=args.frames, n_shots=args.shots, faces=args.faces,
                                 seed=20260925 + i, frame_rate=args.fps)
        ft = v.frames_torch(device)
        tensors.append(ft); videos.append(v)
        clips.append(dict(frames=[ctx.wrap_torch(ft[k]) for k in range(v.n_frames)], times=[v.timestamp(k) for k in range(v.n_frames)],
                          frame_rate=v.frame_rate, shots=v.shots()))
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=args.detect_batch, overlap=not args.no_overlap)
    pipe.return_table = False       # (the float64 host copy of the clustering's table: nobody reads it here)

    def barrier():
        ctx.sync(); torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
    for _ in range(args.warmup):
        pipe.run_many(clips)
    ctx.prof_reset(); ctx.prof_enable(True)
    barrier()
    hbm = HbmSampler(ctx)
    t0 = time.perf_counter()
    results = None
    for _ in range(args.steps):
        results = pipe.run_many(clips)
    barrier()
    elapsed = time.perf_counter() - t0
    hbm.stop()
    ctx.prof_enable(False)
    elapsed = max_over_ranks(elapsed, world, device)
    fam = {}
    for name in FAMILIES:
        ms, n = ctx.prof_get(name)
        fam[name] = {"ms": round(ms, 3), "launches": int(n)}
    if rank != 0:
        return
    total_frames = args.clips * args.frames * args.steps
    fps = total_frames / elapsed
    geo = pipeline.detector_geometry(args.height, args.width)
    flop_per_frame = sum(g[4] for g in geo) * 3100 * 5 * 2.0
    rl = detector_rooflines(fam, args.height, args.width, len(mine) * args.frames * args.steps, args.detect_batch)
    parity, cpu = None, None
    if args.cpu_frames > 0:
        # parity gate: clip 0, a window around its shot cut, product (frames resident) vs the CPU oracle flow
        v, ft = videos[0], tensors[0]
        n = min(16, v.n_frames)
        cut = v.shot_bounds[1] if v.n_shots > 1 else v.n_frames // 2
        i0 = max(0, min(cut - n // 2, v.n_frames - n))
        idx = list(range(i0, i0 + n))
        times = [v.timestamp(i) for i in idx]
        shots = [(a, b) for a, b in v.shots() if b > times[0] and a <= times[-1]]
        res = pipe.run([ctx.wrap_torch(ft[i]) for i in idx], times, v.frame_rate, shots)
        parity, dt, threads = oracle_window_parity([np.ascontiguousarray(ft[i].cpu().numpy()) for i in idx], times, shots, v.frame_rate, v.frame_size, res, lp, ep,
                                                   label="clip 0, frames %d..%d (a window around its cut), product vs CPU oracle flow" % (idx[0], idx[-1]))
        cpu = {"value": round(n / dt, 4), "unit": "frames/s", "cores": int(threads), "kind": "port", "sample": "the same %d-frame 720p window, whole flow" % n}
    if rank == 0 and not args.dense_scoring and not args.small_models:
        # clip 0 (this rank's first) over ALL its frames against the fixture of the CPU oracle flow (tests/golden/c4_clip0.npz)
        fc = full_clip_parity(ctx, clips[0]["frames"], results[0], results[0]["labels"], args, name="c4_clip0", seed=20260925)
        parity = dict(parity or {}, full_clip=fc)
    else:
        fc = None
    out = {"metric": "frames/sec end-to-end detect->embed->cluster, %d independent %dx%d clips farmed one per GPU (BASELINE.json configs[3])" % (args.clips, args.width, args.height),
           "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1000.0 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": DTYPE_NOTE if not args.dense_scoring else DTYPE_NOTE_DENSE,
           "data": "synthetic (seeds 20260925 + clip index)",
           "config": {"workload": "configs[3]: %d clips x %d frames %dx%d %g fps, %d shots and %d faces/frame each, clips round robin over %d GPU(s), frames resident in HBM, "
                                  "per-clip clustering, no collective" % (args.clips, args.frames, args.width, args.height, args.fps, args.shots, args.faces, world),
                      "parallelism": "farm: clip i on rank i %% %d; one engine run per rank (detector of clip i + 1 beside the state machine of clip i)" % world,
                      "detect_batch": args.detect_batch, "collective": "none"},
           "roofline": dict(rl[0], traffic_note="not measured for this configuration (a --pmc pass of its own)"), "roofline_other": rl[1:],
           "screening": ctx.detector_screening_stats() if not args.dense_scoring else None,
           "e2e": e2e_object(flop_per_frame, sum(int(len(r["face_T"])) for r in results) * args.steps, len(mine) * args.frames * args.steps, elapsed),
           "cpu_baseline": cpu, "parity": parity,
           "kernel_families_ms": fam,
           "hbm": hbm.report(frames_bytes=sum(int(t.numel()) for t in tensors)),
           "results": {"clips": len(results), "tracks": sum(len(r["tracks"]) for r in results), "faces_embedded": sum(int(len(r["face_T"])) for r in results),
                       "clusters_per_clip": [len(set(r["labels"].values())) for r in results][:8]},
           "setup_seconds": {"generate_frames_in_hbm": round(t_gen, 1)}}
    out["full_clip_parity"] = full_clip_summary(fc)
    print(json.dumps(out))


class LoopedClips(object):
    """A long video = K differently seeded resident clips played one after the other, again and again (loop j plays clip j % K), delivered
    the way a decoder that writes into HBM would deliver it: every frame is copied into a fresh buffer of the library's (pvf_frame_upload
    from a device address) that the engine releases when it is done with it.  Timestamps continue; the shot cuts repeat with the clips."""

    def __init__(self, ctx, clips_t, n_frames, frame_rate, first_index=0):
        self.ctx, self.clips, self.n, self.frame_rate, self.first = ctx, clips_t, int(n_frames), float(frame_rate), int(first_index)
        self.size = self.frame_size = (int(clips_t[0].shape[2]), int(clips_t[0].shape[1]))

    def __len__(self):
        return self.n

    def source_of(self, i):
        """(clip, frame in the clip) that plays at frame i of the long video"""
        m = int(self.clips[0].shape[0])
        return (i // m) % len(self.clips), i % m

    def __iter__(self):
        h, w = int(self.clips[0].shape[1]), int(self.clips[0].shape[2])
        stride = h * w * 3
        bases = [c.data_ptr() for c in self.clips]
        for i in range(self.n):
            k, j = self.source_of(self.first + i)
            yield (self.first + i) / self.frame_rate, self.ctx.upload_device(bases[k] + j * stride, h, w, transient=True)


def bench_stream(args, rank, local_rank, world, device, lp, ep):
    """BASELINE.json configs[2]: one long 1080p video cut into frame ranges at shot boundaries, one range per GPU, streamed through the
    bounded-memory engine (FacePipeline.run_stream): frames arrive one by one, shots are detected / tracked / extracted in flight and
    their frames released, then ONE all-gather of the (128 float32, time, track) rows and one global clustering."""
    import numpy as np
    import torch
    from pyannote_video_amd import synth, pipeline, dist as pdist
    from pyannote_video_amd.runtime import Context
    clip_n = 1000
    n = args.frames
    first = rank * n                                        # this rank's frame range of the world * n frame video
    loops = range(first // clip_n, (first + n - 1) // clip_n + 1)
    K = max(1, min(int(args.distinct_clips), len(loops)))
    # the video plays K differently seeded clips (identities from a pool of 250) one after the other: loop j plays clip j % K.  Round 3
    # looped ONE clip, so the 720 tracks of a 22 500-frame range were 22.5 copies of the same 32.
    videos = [synth.SyntheticVideo(width=args.width, height=args.height, n_frames=clip_n, n_shots=args.shots, faces=args.faces, identities=250,
                                   seed=20260925 + k, frame_rate=args.fps) for k in range(K)]
    t_gen = time.time()
    clips_t = [v.frames_torch(device) for v in videos]
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    ctx = Context(device=local_rank)
    if args.dense_scoring:
        ctx.detector_screening(False)
    pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=args.detect_batch, overlap=not args.no_overlap)
    pipe.return_table = False       # (the float64 host copy of the clustering's table: nobody reads it here)
    per_shot = clip_n // args.shots
    assert n % per_shot == 0, "--frames must be a multiple of the shot length (%d) so that the ranges are cut at shot boundaries" % per_shot
    shots = [((first + k * per_shot) / args.fps, (first + (k + 1) * per_shot) / args.fps) for k in range(n // per_shot)]

    def step(frames=n):
        tm = {}
        t_step = time.perf_counter()
        src = LoopedClips(ctx, clips_t, frames, args.fps, first_index=first)
        res = pipe.run_stream(src, shots[:frames // per_shot], timings=tm, cluster=False, last_shard=(rank == world - 1), reorder=(world == 1))
        T, ids, X, offsets = pdist.gather_rows(res["face_T"], res["face_id"], res["embeddings"], len(res["tracks"]), device=device,
                                               file_T=res["file_T"] if world > 1 else None, file_id=res["file_id"] if world > 1 else None)
        t0 = time.perf_counter()
        labels = pdist.global_cluster(pipe.clustering, T, ids, X)
        tm["cluster_s"] = time.perf_counter() - t0
        tm["step_wall_s"] = time.perf_counter() - t_step
        return res, labels, tm

    def barrier():
        ctx.sync(); torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
    for _ in range(args.warmup):
        step(min(n, 4 * per_shot))                          # plans, scratch buffers and the buffer pool exist after four shots
    ctx.prof_reset(); ctx.prof_enable(True)
    barrier()
    hbm = HbmSampler(ctx)
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    barrier()
    elapsed = time.perf_counter() - t0
    hbm.stop()
    ctx.prof_enable(False)
    elapsed = max_over_ranks(elapsed, world, device)
    res, labels, tm = last
    fam = {}
    for name in FAMILIES:
        ms, k = ctx.prof_get(name)
        fam[name] = {"ms": round(ms, 3), "launches": int(k)}
    if rank != 0:
        return
    fps = n * world * args.steps / elapsed
    geo = pipeline.detector_geometry(args.height, args.width)
    flop_per_frame = sum(g[4] for g in geo) * 3100 * 5 * 2.0
    rl = detector_rooflines(fam, args.height, args.width, n * args.steps, args.detect_batch)
    parity, cpu, cluster_check = None, None, None
    if world == 1 and args.cpu_frames > 0:
        # parity gate through the STREAMING path: a window across the seam between two loops (last frames of one clip's last shot, first
        # frames of the next clip's first shot: a cut), product (run_stream, frames delivered one by one) vs the CPU oracle flow
        m = min(8, args.cpu_frames)
        i0 = clip_n - m // 2
        idx = list(range(i0, i0 + m))
        times = [i / args.fps for i in idx]
        wshots = [(a, b) for a, b in shots if b > times[0] and a <= times[-1]]
        win = LoopedClips(ctx, clips_t, m, args.fps, first_index=i0)
        r = pipe.run_stream(win, wshots)
        frames_np = [np.ascontiguousarray(clips_t[k][j].cpu().numpy()) for k, j in (win.source_of(i) for i in idx)]
        parity, dt, threads = oracle_window_parity(frames_np, times, wshots, args.fps, videos[0].frame_size, r, lp, ep,
                                                   label="frames %d..%d of the long video (across the seam between two clips: a cut), streamed product vs CPU oracle flow" % (idx[0], idx[-1]))
        cpu = {"value": round(m / dt, 4), "unit": "frames/s", "cores": int(threads), "kind": "port", "sample": "the same %d-frame window, whole flow" % m}
    if world == 1 and args.cluster_check_frames != 0 and len(res["face_T"]):
        cluster_check = oracle_cluster_check(pipe, res, args.cluster_check_frames / args.fps if args.cluster_check_frames > 0 else float("inf"))
    full_clip = None
    if rank == 0 and first == 0 and n >= clip_n and not args.dense_scoring and not args.small_models:
        # the first 1000-frame clip of the long video, as the STREAMED run produced it, against the oracle flow's output over all its frames
        # (tests/golden/c3_clip0.npz); the raw-candidate pass runs on the clip's resident frames
        clip_frames = [ctx.wrap_torch(clips_t[0][i]) for i in range(clip_n)]
        full_clip = full_clip_parity(ctx, clip_frames, res, None, args, name="c3_clip0", identities=250, n_frames=clip_n)
        parity = dict(parity or {}, full_clip=full_clip)
    peak_frames = res.get("peak_frames_resident")
    idents = set(tr["ident"] for v in videos for shot in v.tracks for tr in shot)       # identity k looks the same in every clip
    out = {"metric": "frames/sec end-to-end detect->embed->cluster, one long 1080p@25fps video in frame ranges (BASELINE.json configs[2])",
           "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1000.0 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": DTYPE_NOTE if not args.dense_scoring else DTYPE_NOTE_DENSE,
           "data": "synthetic: %d differently seeded 1000-frame clips (identities from a pool of 250) played one after the other, loop j = clip j %% %d; timestamps and "
                   "track numbers continue; every loop adds its tracks to the global clustering" % (K, K),
           "config": {"workload": "configs[2]: %d frames per GPU (%.1f min of 1080p 25 fps video; %d GPUs x that = the whole video), %d-frame shots, %d faces/frame; frames delivered one by one "
                                  "into HBM buffers of the library (device-to-device from the resident clips), released shot by shot" % (n, n / args.fps / 60.0, world, per_shot, args.faces),
                      "parallelism": "frame ranges cut at shot boundaries x%d + all-gather of track embeddings + one global clustering" % world if world > 1 else "single GPU (one range)",
                      "detect_batch": args.detect_batch, "collective": pdist.collective_name(), "distinct_clips": K},
           "roofline": dict(rl[0], traffic_note="not measured for this configuration (a --pmc pass of its own)"), "roofline_other": rl[1:],
           "screening": ctx.detector_screening_stats() if not args.dense_scoring else None,
           "e2e": e2e_object(flop_per_frame, int(len(res["face_T"])) * args.steps, n * args.steps, elapsed),
           "cpu_baseline": cpu, "parity": parity, "cluster_check": cluster_check,
           "stage_seconds_last_step": {k: round(v, 3) for k, v in tm.items()},
           "kernel_families_ms": fam,
           "hbm": dict(hbm.report(frames_bytes=sum(int(c.numel()) for c in clips_t)), peak_frames_resident=peak_frames,
                       whole_range_resident_would_be_bytes=int(n) * args.width * args.height * 3),
           "results": {"tracks": len(res["tracks"]), "faces_embedded": int(len(res["face_T"])), "clusters": len(set(labels.values())),
                       "tracks_clustered_globally": len(labels), "labels_sha256_16": labels_digest(labels),
                       "identities_in_video": len(idents)},
           "setup_seconds": {"generate_clips_in_hbm": round(t_gen, 1)}}
    out["full_clip_parity"] = full_clip_summary(full_clip)
    print(json.dumps(out))


def oracle_cluster_check(pipe, res, t_end):
    """The clustering kernels against the CPU oracle on the pipeline's own descriptors: the tracks that END before t_end seconds (long
    tracks of ~250 rows each, real embeddings) -- pair means + agglomeration by the oracle (C, all cores) and by the product on the same
    rows; labels and merge order must agree, D to 1e-12."""
    import numpy as np
    from oracle import oracle
    from pyannote_video_amd import _lib
    T_, ids_, emb = np.asarray(res["face_T"]), np.asarray(res["face_id"]), res["embeddings"]
    last = {}
    for t, i in zip(T_.tolist(), ids_.tolist()):
        last[i] = max(last.get(i, t), t)
    keep_ids = sorted(i for i, t in last.items() if t < t_end)
    sel = np.isin(ids_, keep_ids)
    if sel.sum() < 2:
        return None
    tids, order, row_start = pipe.clustering.plan_rows(T_[sel], ids_[sel])
    if len(tids) < 2:
        return None
    E = np.ascontiguousarray(emb[sel])
    X = _lib.round_rows(E, 5)[order]
    oracle.lib().pvo_set_threads(oracle.usable_cpus(cap=1024))
    t0 = time.perf_counter()
    Dr = oracle.pair_mean_dist(X, row_start)
    lr, logr = oracle.hac(Dr, np.diff(row_start), pipe.clustering.threshold)
    dt = time.perf_counter() - t0
    ctx = pipe.ctx
    D = ctx.pair_mean_dist(X, row_start)
    lg, logg = ctx.cluster_tracks_f32(E, order, row_start, pipe.clustering.threshold)
    rel = float(np.max(np.abs(D - Dr) / np.maximum(Dr, 1e-300))) if len(tids) > 1 else 0.0
    if os.environ.get("PVF_DUMP_MERGE_LOGS"):                 # the two merge logs, for a look at a verdict off line
        np.savez(os.environ["PVF_DUMP_MERGE_LOGS"], product=logg, oracle=logr)
    return {"tracks": int(len(tids)), "rows": int(row_start[-1]), "oracle_seconds": round(dt, 2),
            "labels": "exact" if np.array_equal(lg, lr) else "MISMATCH",
            "merge_order": merge_order_verdict(logg, logr),
            "D_max_rel_err": rel, "clusters": int(len(set(lr.tolist())))}


def merge_order_verdict(logg, logr, tol=1e-10):
    """the product's merge log (a, b, distance, new size: cluster b joins cluster a) against the oracle's: "exact" when the same pairs
    merge in the same order.  The long video replays its clips: a track and its replays have the same descriptors, so they are at exactly
    the same distance from everything else -- the oracle breaks such ties by position (first minimum in row-major order), the product's
    tiled sums break them by their last bits (the two tables agree to 6e-14 relative), and the ORDER in which interchangeable tracks join a
    cluster -- and with it the pair ids, even the sizes on the way (four replicas pair up two and two, or one after the other) -- is not
    defined by the algorithm.  What is defined, and compared here: the merge distance at every step (`tol` relative), the partition of the
    tracks wherever the two orders have caught up with each other (a stretch of steps inside which the partitions differ must end in
    the SAME partition), the sizes outside such stretches, and (by the caller) the final labels: the dendrogram up to its ties."""
    import numpy as np
    if len(logg) != len(logr):
        return "MISMATCH (%d vs %d merges)" % (len(logg), len(logr))
    n = len(logr)
    if n == 0:
        return "exact"
    close = np.abs(logg[:, 2] - logr[:, 2]) <= tol * np.maximum(np.abs(logr[:, 2]), 1e-300)
    if not close.all():
        return "MISMATCH at merge %d (distance %.12g against the oracle's %.12g)" % (int(np.argmin(close)), logg[np.argmin(close), 2], logr[np.argmin(close), 2])
    if np.array_equal(logg[:, :2], logr[:, :2]) and np.array_equal(logg[:, 3], logr[:, 3]):
        return "exact"
    m = int(max(logg[:, :2].max(), logr[:, :2].max())) + 1

    def canon(lab):                                        # every cluster named by its smallest member
        order = np.argsort(lab, kind="stable")
        first = np.ones(m, bool); first[1:] = lab[order][1:] != lab[order][:-1]
        out = np.zeros(m, np.int64)
        out[order] = order[np.maximum.accumulate(np.where(first, np.arange(m), 0))]
        return out

    lab_g, lab_r = np.arange(m), np.arange(m)
    stretches, longest, run, agree_before = 0, 0, 0, True
    for i in range(n):
        for lab, log in ((lab_g, logg), (lab_r, logr)):
            a, b = int(log[i, 0]), int(log[i, 1])
            lab[lab == lab[b]] = lab[a]
        agree = np.array_equal(canon(lab_g), canon(lab_r))
        if agree and agree_before and logg[i, 3] != logr[i, 3]:
            return "MISMATCH at merge %d (cluster size %d against the oracle's %d)" % (i, int(logg[i, 3]), int(logr[i, 3]))
        if not agree:
            run += 1
            if agree_before:
                stretches += 1
            longest = max(longest, run)
        else:
            run = 0
        agree_before = agree
    if not agree_before:
        return "MISMATCH: the partitions differ from merge %d to the end" % (n - run)
    first = int(np.nonzero((logg[:, :2] != logr[:, :2]).any(axis=1))[0][0]) if not np.array_equal(logg[:, :2], logr[:, :2]) else -1
    return ("equal up to ties: the same merge distance (%.0e relative) at every one of the %d merges; the partitions of the tracks coincide except inside %d "
            "stretch(es) of at most %d merges in which interchangeable tracks (replayed clips: identical descriptors) join a cluster in another order, "
            "each ending in the same partition; first differing pair at merge %d" % (tol, n, stretches, longest, first))


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
        tracks = ref_flow.track_video(frames, times, shots, det, lambda: oracle.Tracker(tabs), video.frame_rate, detect_every=args.detect_every,
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
           "sample": "frames %d..%d of the same video (a window around the shot cut at frame %d), whole flow: detect + fwd/bwd tracking %.1f s "
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
        "sample": "product pipeline vs CPU oracle flow on frames %d..%d (%dx%d, full landmark model, detect batch %d)" % (idx[0], idx[-1], args.width, args.height, args.detect_batch),
        "boxes": "exact" if res["tracks"] == tracks else "MISMATCH",           # every track row: time, detector / tracker box, status string
        "track_ids": "exact" if [len(t) for t in res["tracks"]] == [len(t) for t in tracks] and same_rows else "MISMATCH",
        "landmarks": "exact" if same_rows and np.array_equal(res["landmarks"].astype(np.int64), ref_int) else "MISMATCH",
        "embed_l2_max": float(np.linalg.norm(ref_e - res["embeddings"].astype(np.float64), axis=1).max()) if same_rows and len(ref_e) else None,
        "labels": "exact" if res["labels"] == labels else "MISMATCH",
        "tracks": len(tracks), "faces": int(len(ref_e)),
    }
    if args.detect_every == 0.0 and video.n_frames >= 64:
        # a window nobody chose: 8 frames at a position drawn from a seed that changes with every run and is printed (the
        # other windows always sit on a cut)
        seed = args.parity_seed if args.parity_seed is not None else int(time.time())
        i0 = int(np.random.default_rng(seed).integers(0, video.n_frames - 8))
        idx3 = list(range(i0, i0 + 8))
        times3 = [video.timestamp(i) for i in idx3]
        shots3 = [(a, b) for a, b in video.shots() if b > times3[0] and a <= times3[-1]]
        res3 = pipe.run([ctx.wrap_torch(frames_t[i]) for i in idx3], times3, video.frame_rate, shots3)
        p3, _, _ = oracle_window_parity([np.ascontiguousarray(frames_t[i].cpu().numpy()) for i in idx3], times3, shots3, video.frame_rate, video.frame_size, res3, lp, ep,
                                        label="frames %d..%d (drawn from --parity-seed %d)" % (idx3[0], idx3[-1], seed))
        p3["seed"] = seed
        parity["random_window"] = p3
    if video.n_shots >= 4 and args.config == "c2" and args.detect_every == 0.0:
        # a second, smaller window at the LAST cut of the clip (shots 3 | 4): other faces, other backgrounds, other tracker histories
        cut2 = video.shot_bounds[video.n_shots - 1]
        idx2 = list(range(cut2 - 4, cut2 + 4))
        times2 = [video.timestamp(i) for i in idx2]
        shots2 = [(a, b) for a, b in video.shots() if b > times2[0] and a <= times2[-1]]
        res2 = pipe.run([ctx.wrap_torch(frames_t[i]) for i in idx2], times2, video.frame_rate, shots2)
        p2, _, _ = oracle_window_parity([np.ascontiguousarray(frames_t[i].cpu().numpy()) for i in idx2], times2, shots2, video.frame_rate, video.frame_size, res2, lp, ep,
                                        label="frames %d..%d (the cut between the last two shots)" % (idx2[0], idx2[-1]))
        parity["second_window"] = p2
    return cpu, parity


if __name__ == "__main__":
    main()
