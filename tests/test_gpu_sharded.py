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
T, ids, X, offsets = pd.gather_rows(res["face_T"], res["face_id"], res["X"], len(res["tracks"]), file_T=res["file_T"], file_id=res["file_id"])
labels = pd.global_cluster(pipe.clustering, T, ids, X)
out = {"rank": rank, "n_tracks": len(res["tracks"]), "offsets": offsets, "T": T.tolist(), "ids": ids.tolist(),
       "Xsum": float(np.abs(X).sum()), "labels": sorted(labels.items())}
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
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
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
    assert abs(r0["Xsum"] - float(np.abs(res["X"]).sum())) < 1e-9
    assert dict(map(tuple, r0["labels"])) == res["labels"]
    ctx.close()


def test_rccl_allgather_rows_single_rank():
    """libpvface_dist.so on the one GPU of this box: communicator of size 1 -- id, ncclCommInitRank, both collectives of a gather, teardown
    (a box with one GPU cannot host two ranks of one RCCL communicator; the N > 1 exchange logic is covered over gloo on CPU)"""
    from pyannote_video_amd import dist
    rng = np.random.default_rng(3)
    comm = dist.RcclRows(0, 0, 1, dist.RcclRows.unique_id())
    for n in (0, 1, 37, 5000):
        rows = rng.normal(size=(n, 131))
        out, counts = comm.allgather_rows(rows)
        assert counts == [n] and out.shape == (n, 131) and np.array_equal(out, rows)
    D = rng.normal(size=(40, 40))
    out, counts = comm.allgather_rows(D)
    assert np.array_equal(out, D)
    comm.close()


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
