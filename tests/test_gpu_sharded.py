"""GPU: the multi-GPU decomposition on one device.  Two processes (gloo rendezvous on 127.0.0.1, both on cuda:0) each run
detect -> track -> extract on their shot range of one clip, exchange embedding rows with dist.gather_rows and cluster
globally; the result must equal the single-process run over the whole clip: same track ids (local id + prefix offset),
same rows, same labels.  Also the config-5 style clustering stress at reduced scale."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "pyannote-video_amd"))
import torch.distributed as dist
from pyannote_video_amd import synth, models, pipeline, dist as pd
from pyannote_video_amd.runtime import Context
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
v = synth.SyntheticVideo(width=480, height=270, n_frames=16, n_shots=4, faces=2, min_face=50, max_face=90, seed=5)
lp, ep = models.ensure_synthetic_models(sys.argv[3], small=True)
ctx = Context(device=0)
pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=4)
times = [v.timestamp(i) for i in range(v.n_frames)]
ranges = pipeline.split_into_shots(times, v.shots())
s0, s1 = pd.shard_shots(ranges, world)[rank]
i0, i1 = ranges[s0][0], ranges[s1 - 1][1]
frames = [ctx.upload(v.frame(i)) for i in range(i0, i1)]
res = pipe.run(frames, times[i0:i1], v.frame_rate, v.shots()[s0:s1], cluster=False, last_shard=(rank == world - 1), reorder=False)
T, ids, X, offsets = pd.gather_rows(res["face_T"], res["face_id"], res["embeddings"], len(res["tracks"]), file_T=res["file_T"], file_id=res["file_id"])
labels = pd.global_cluster(pipe.clustering, T, ids, X)          # split over the two ranks: upper-triangle shares, gathered, mirrored
out = {"rank": rank, "n_tracks": len(res["tracks"]), "offsets": offsets, "T": T.tolist(), "ids": ids.tolist(), "in_hbm": hasattr(X.rows, "ptr"),
       "Xsum": float(np.abs(X.numpy().astype(np.float64)).sum()), "labels": sorted(labels.items()), "history": pipe.clustering.history}
open(sys.argv[2] + ".%d" % rank, "w").write(json.dumps(out))
dist.barrier(); dist.destroy_process_group(); ctx.close()
'''


def test_two_shards_equal_single_process(tmp_path, model_dir):
    from pyannote_video_amd import synth, pipeline
    from pyannote_video_amd.runtime import Context
    from pyannote_video_amd import models
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "out")
    # two ranks on ONE device cannot form an RCCL communicator: gloo rendezvous + the torch.distributed collectives, asked for explicitly
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PVF_DIST_COLLECTIVE="torch")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                           "--master-port", "29633", str(script), ROOT, out, model_dir], env=env, timeout=600)
    r0, r1 = (json.loads(open(out + ".%d" % r).read()) for r in (0, 1))
    # single process over the whole clip
    v = synth.SyntheticVideo(width=480, height=270, n_frames=16, n_shots=4, faces=2, min_face=50, max_face=90, seed=5)
    lp, ep = models.ensure_synthetic_models(model_dir, small=True)
    ctx = Context(device=0)
    pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=4, overlap=False)
    times = [v.timestamp(i) for i in range(v.n_frames)]
    res = pipe.run([ctx.upload(v.frame(i)) for i in range(v.n_frames)], times, v.frame_rate, v.shots())
    assert r0["offsets"] == r1["offsets"] == [0, r0["n_tracks"]]
    assert r0["n_tracks"] + r1["n_tracks"] == len(res["tracks"])
    assert r0["ids"] == r1["ids"] and r0["labels"] == r1["labels"]
    # only the shard that ends the video applies getFaceGenerator's dropped-last-group habit => identical rows, ids, labels
    assert r0["T"] == res["face_T"].tolist() and r0["ids"] == res["face_id"].tolist()
    assert r0["in_hbm"] and r1["in_hbm"]                 # the gathered descriptors stay in device memory
    assert abs(r0["Xsum"] - float(np.abs(res["embeddings"].astype(np.float64)).sum())) < 1e-9
    assert dict(map(tuple, r0["labels"])) == res["labels"]
    assert [tuple(h) for h in r0["history"]] == [tuple(h) for h in r1["history"]] == pipe.clustering.history     # same merges, same distances
    ctx.close()


RCCL_WORKER = r'''
import os, sys
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "pyannote-video_amd"))
import torch                                   # FIRST, as in bench.py: the process then runs on the HIP runtime and the RCCL torch ships
import torch.distributed as td
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
td.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(4, device="cuda"); td.all_reduce(x)               # torch's own RCCL communicator is live beside ours
from pyannote_video_amd import dist
from pyannote_video_amd.runtime import Context, DeviceRows
import numpy as np
comm = dist.RcclRows(0, 0, 1, dist.RcclRows.unique_id())
assert comm.counts(37) == [37]
g = torch.Generator().manual_seed(3)
for n, k in ((0, 528), (1, 528), (37, 528), (5000, 528), (40, 320)):
    rows = torch.randint(0, 256, (n, k), dtype=torch.uint8, generator=g).cuda()
    out, counts = comm.allgather(rows)
    assert counts == [n] and tuple(out.shape) == (n, k) and out.is_cuda and torch.equal(out, rows)
    assert n == 0 or out.data_ptr() != rows.data_ptr()
# the whole device-resident exchange + split clustering with this communicator as the job's exchange step (world 1: one share)
dist._exchange["tried"] = True; dist._exchange["comm"] = comm
rng = np.random.default_rng(5)
T, ids = np.repeat(np.arange(12) * 1.0, 5) + np.tile(np.arange(5) * 0.04, 12), np.repeat(np.arange(12), 5)
cent = rng.normal(size=(3, 128)); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
E = (0.55 * cent[ids % 3] + 0.01 * rng.normal(size=(60, 128))).astype(np.float32)
gT, gid, X, off = dist.gather_rows(T, ids, E, 12, file_T=T, file_id=ids)
assert off == [0] and isinstance(X.rows, DeviceRows) and np.array_equal(X.numpy(), E[X.index]) and np.array_equal(gT, T[X.index])
from pyannote_video_amd.clustering import FaceClustering
ctx = Context(device=0, detector=None)
fc = FaceClustering(ctx=ctx); fc.shard = dist.DistanceShard(0, 1)
lab_split = fc.cluster_rows(gT, gid, X.rows, src_index=X.index)      # rows in HBM -> upper rows -> all-gather (RCCL) -> mirror + HAC
lab_one = FaceClustering(ctx=ctx).cluster_rows(T, ids, E)
assert lab_split == lab_one and len(set(lab_one.values())) == 3
comm.close(); ctx.close(); td.destroy_process_group()
print("RCCL-OK")
'''


def test_rccl_allgatherv_single_rank(tmp_path):
    """libpvface_dist.so on the one GPU of this box, in a process set up like bench.py's ranks (torch imported first, the job's nccl
    process group alive): communicator of size 1 -- id, ncclCommInitRank, the count exchange, the grouped ncclBroadcast all-gather
    between DEVICE buffers -- then gather_rows + the split clustering through it with every payload in HBM, and teardown.  (A box with
    one GPU cannot host two ranks of one RCCL communicator; the N > 1 exchange logic is covered over gloo on CPU and with two processes
    on this GPU above.)"""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    env.pop("PVF_DIST_COLLECTIVE", None)
    p = subprocess.run([sys.executable, str(script), ROOT, "29655"], env=env, timeout=600, capture_output=True, text=True)
    assert p.returncode == 0 and "RCCL-OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_bench_launches_ranks_itself_and_two_ranks_equal_one(tmp_path):
    """`python bench.py --gpus 2` starts two ranks by itself (the flag used to be parsed and ignored).  On a 1-GPU box
    --oversubscribe puts both on device 0 (gloo rendezvous, torch.distributed collectives): launcher, shot-range sharding, the 528-byte
    row exchange, the split upper-triangle distances and the global clustering all run; the labels equal the 1-rank run on the same 128
    frames.  Without --oversubscribe the same command is an error here, not a silent 1-GPU measurement."""
    common = ["--frames", "128", "--steps", "1", "--warmup", "0", "--cpu-frames", "0", "--no-dropin", "--no-host-ingest", "--scaling", "strong",
              "--detect-batch", "32"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PVF_DIST_COLLECTIVE", None)

    def run(extra):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, env=env, timeout=900, capture_output=True, text=True)
        return p
    p2 = run(["--gpus", "2", "--oversubscribe"])
    assert p2.returncode == 0, p2.stderr[-3000:]
    l2 = json.loads([l for l in p2.stdout.splitlines() if l.startswith("{")][-1])
    p1 = run(["--gpus", "1"])
    assert p1.returncode == 0, p1.stderr[-3000:]
    l1 = json.loads([l for l in p1.stdout.splitlines() if l.startswith("{")][-1])
    assert l2["n_gpus"] == 2 and l1["n_gpus"] == 1
    assert l2["config"]["oversubscribed"] is True and l2["config"]["devices"] == 1 and l2["config"]["collective"] == "torch"
    assert l1["config"]["collective"] == "none" and l2["scaling"] == "strong"
    assert l2["results"]["tracks_clustered_globally"] == l1["results"]["tracks_clustered_globally"] > 0
    assert l2["results"]["labels_sha256_16"] == l1["results"]["labels_sha256_16"]
    assert l2["results"]["clusters"] == l1["results"]["clusters"]
    from pyannote_video_amd import _lib
    if _lib.device_count() < 2:
        p = run(["--gpus", "2"])
        assert p.returncode != 0 and "GPU" in p.stderr


def test_eight_ranks_through_the_launcher_equal_one(tmp_path):
    """The shape of the first 8-GPU run, on one GPU: `bench.py --gpus 8 --oversubscribe --scaling strong` on one 2000-frame video of 8
    shots (960 x 540, so that eight contexts fit one device) -- eight ranks take one shot each, exchange their rows, split the distance
    step by triangle area (8 shares) and cluster globally; the labels digest equals the 1-rank run's, and the line carries every rank's own
    time and where its last step went."""
    common = ["--frames", "2000", "--shots", "8", "--width", "960", "--height", "540", "--steps", "1", "--warmup", "0", "--cpu-frames", "0",
              "--no-dropin", "--no-host-ingest", "--no-dense-leg", "--scaling", "strong", "--detect-batch", "32"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    env.pop("PVF_DIST_COLLECTIVE", None)

    def run(extra):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, env=env, timeout=1500, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    l8 = run(["--gpus", "8", "--oversubscribe"])
    l1 = run(["--gpus", "1"])
    assert l8["n_gpus"] == 8 and l8["config"]["oversubscribed"] is True and l8["scaling"] == "strong"
    assert l8["results"]["tracks_clustered_globally"] == l1["results"]["tracks_clustered_globally"] > 0
    assert l8["results"]["labels_sha256_16"] == l1["results"]["labels_sha256_16"]
    assert l8["results"]["clusters"] == l1["results"]["clusters"]
    pr = l8["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8)) and sum(r["frames"] for r in pr) == 2000
    assert sum(r["tracks"] for r in pr) == l1["results"]["tracks"] and sum(r["faces"] for r in pr) == l1["results"]["faces_embedded"]
    assert all("exchange_s" in r["last_step_s"] and "cluster_s" in r["last_step_s"] for r in pr)
    assert l1["per_rank"] is None


def test_cluster_stress_reduced_config5(ctx, oracle):
    """config 5 at reduced scale: T = 1500 tracks x 4 rows around 120 centres, distances bracket the 0.6 threshold"""
    rng = np.random.default_rng(17)
    K, T, R = 120, 1500, 4
    cent = rng.normal(size=(K, 128)); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    ident = rng.integers(0, K, T)
    X = np.concatenate([np.round(0.55 * (lambda x: x / np.linalg.norm(x, axis=1, keepdims=True))(cent[ident[t]] + 0.05 * rng.normal(size=(R, 128))), 5)
                        for t in range(T)])
    rs = (np.arange(T + 1) * R).astype(np.int32)
    labels, log = ctx.cluster_tracks(X, rs, 0.6)
    assert len(log) == T - len(set(labels.tolist()))
    # every cluster is pure and every identity is one cluster (well separated data)
    for t in range(T):
        assert ident[labels[t]] == ident[t]
    assert len(set(labels.tolist())) == len(set(ident.tolist()))
    # first merges agree with the CPU oracle on a sub-problem
    sub = 200
    Ds = oracle.pair_mean_dist(X[:sub * R], rs[:sub + 1])
    lo, _ = oracle.hac(Ds, [R] * sub, 0.6)
    lg, _ = ctx.cluster_tracks(X[:sub * R], rs[:sub + 1], 0.6)
    assert np.array_equal(lo, lg)


STALL_WORKER = r'''
import os, sys, time
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "pyannote-video_amd"))
import torch
from pyannote_video_amd import dist
comm = dist.RcclRows(0, 0, 1, dist.RcclRows.unique_id())
os.environ["PVF_DIST_TIMEOUT_S"] = "5"
comm.stall(200)                                  # a collective that takes 0.2 s: well inside the limit
assert comm.counts(7) == [7]
os.environ["PVF_DIST_TIMEOUT_S"] = "1"
t0 = time.time()
try:
    comm.stall(4000)                             # "a peer that never joins": 4 s against a limit of 1 s
    print("NO-ERROR")
except RuntimeError as e:
    msg = str(e)
    print("ERR:", msg)
    took = time.time() - t0
    ok = ("rank 0 of 1" in msg and "gave up after" in msg and "PVF_DIST_TIMEOUT_S" in msg and "spins for 4000 ms" in msg and "aborted" in msg)
    assert ok, msg
    assert 0.9 < took < 16.0, took                # gave up at the limit, then drained the stream (the spin ends by itself after 4 s)
    try:
        comm.counts(1)
        print("NO-SECOND-ERROR")
    except RuntimeError as e2:
        assert "aborted by an earlier failure" in str(e2), str(e2)
        comm.close()
        print("WATCHDOG-OK")
'''


def test_watchdog_gives_up_on_a_collective_that_never_completes(tmp_path):
    """csrc/dist.hip wait_collective: a spinning kernel stands where a collective whose peer never joins would stand (pvfd_debug_stall).
    Inside the limit the call returns; beyond it the rank gives up after PVF_DIST_TIMEOUT_S, aborts the communicator, drains the stream
    and fails with a message naming the rank, the world size and what it was waiting for; the handle then refuses further collectives
    and can still be destroyed."""
    script = tmp_path / "stall_worker.py"
    script.write_text(STALL_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, timeout=300, capture_output=True, text=True)
    marks = [l for l in p.stdout.splitlines() if l.startswith(("ERR:", "NO-", "WATCHDOG-OK"))]
    assert p.returncode == 0 and "WATCHDOG-OK" in marks, (marks, p.stdout[-800:], p.stderr[-2000:])
