"""CPU, world_size 2 over gloo: the one exchange step of the multi-GPU path (all-gather of per-rank embedding rows with
global track-id offsets) and the shard planner.  Rendezvous on 127.0.0.1."""
import os
import sys
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "pyannote-video_amd"))
import torch.distributed as dist
from pyannote_video_amd import dist as pd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
if os.environ.get("PVF_TEST_EXPECT_REFUSAL"):
    # a job over gloo that did not ask for the torch.distributed collectives: the exchange must refuse, not pick one silently
    try:
        pd.exchange()
        out = {"rank": rank, "refused": False}
    except RuntimeError as e:
        out = {"rank": rank, "refused": "PVF_DIST_COLLECTIVE=torch" in str(e)}
    open(sys.argv[2] + ".%d" % rank, "w").write(json.dumps(out))
    dist.barrier(); dist.destroy_process_group(); sys.exit(0)
rng = np.random.default_rng(100 + rank)
n_tracks = 3 + rank
rows = 5 + 2 * rank
T = rng.random(rows) + 10 * rank
ids = rng.integers(0, n_tracks, rows)
X = rng.normal(size=(rows, 128)).astype(np.float32)
gT, gid, gX, offsets = pd.gather_rows(T, ids, X, n_tracks)
gX = gX.numpy()
out = {"rank": rank, "n": int(len(gT)), "offsets": offsets, "ids": gid.tolist(), "sumX": float(gX.astype(np.float64).sum()), "T0": float(gT[0]), "Tlast": float(gT[-1]),
       "collective": pd.collective_name(), "mine_back": bool(np.array_equal(gX[sum(5 + 2 * r for r in range(rank)):][:rows], X))}
open(sys.argv[2] + ".%d" % rank, "w").write(json.dumps(out))
dist.barrier(); dist.destroy_process_group()
'''


import pytest


def _run2(script, args, port, **env):
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), str(script)] + args, env=dict(os.environ, MASTER_ADDR="127.0.0.1", **env), timeout=240)


def test_gather_rows_world2(tmp_path):
    """528-byte rows (float32[128], float64 time, int32 local id) gathered over gloo (PVF_DIST_COLLECTIVE=torch, the explicit opt-in)"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "out")
    _run2(script, [ROOT, out], 29613, PVF_DIST_COLLECTIVE="torch")
    import json
    r0, r1 = (json.loads(open(out + ".%d" % r).read()) for r in (0, 1))
    assert r0["n"] == r1["n"] == 5 + 7
    assert r0["collective"] == r1["collective"] == "torch"
    assert r0["offsets"] == r1["offsets"] == [0, 3]
    assert r0["ids"] == r1["ids"]                       # every rank sees the same global rows, rank order
    assert max(r0["ids"][:5]) < 3 and min(r0["ids"][5:]) >= 3
    assert abs(r0["sumX"] - r1["sumX"]) < 1e-9
    assert r0["mine_back"] and r1["mine_back"]          # the float32 values travel unchanged
    rng0, rng1 = np.random.default_rng(100), np.random.default_rng(101)
    t0 = rng0.random(5); t1 = rng1.random(7) + 10
    assert r0["T0"] == t0[0] and r0["Tlast"] == t1[-1]


def test_exchange_refuses_to_pick_a_collective_silently(tmp_path):
    """ONE collective path: over a backend where libpvface_dist.so's RCCL communicator cannot exist (gloo) the exchange raises unless
    PVF_DIST_COLLECTIVE=torch asked for the torch.distributed collectives (no silent dual back-end)"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "out")
    env = {k: v for k, v in os.environ.items() if k != "PVF_DIST_COLLECTIVE"}
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                           "--master-port", "29615", str(script), ROOT, out], env=dict(env, MASTER_ADDR="127.0.0.1", PVF_TEST_EXPECT_REFUSAL="1"), timeout=240)
    import json
    assert all(json.loads(open(out + ".%d" % r).read())["refused"] is True for r in (0, 1))


ORDER_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "pyannote-video_amd"))
import torch.distributed as dist
from pyannote_video_amd import dist as pd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
table = json.loads(open(sys.argv[3]).read())[rank]                     # this rank's share of the track table, file order
fT, fid = np.array(table["T"]), np.array(table["id"])
order = np.random.default_rng(rank).permutation(len(fT))               # faces arrive in any order within the shard
X = np.zeros((len(fT), 128), np.float32); X[:, 0] = (fT[order] * 1000).round(); X[:, 1] = fid[order]      # (float32-exact keys)
gT, gid, gX, offsets = pd.gather_rows(fT[order], fid[order], X, table["n_tracks"], file_T=fT, file_id=fid)
open(sys.argv[2] + ".%d" % rank, "w").write(json.dumps({"T": gT.tolist(), "ids": gid.tolist(), "x0": gX.numpy()[:, 0].tolist()}))
dist.barrier(); dist.destroy_process_group()
'''


def test_gather_rows_restores_reference_row_order_world2(tmp_path):
    """faces of one timestamp come out in the order pandas' (unstable) sort of the WHOLE track table gives (formats.file_order),
    however the table was split over the ranks"""
    import json
    from pyannote_video_amd import formats
    rng = np.random.default_rng(5)
    shares, fT, fid, base = [], [], [], 0
    for rank, (f0, f1) in enumerate([(0, 40), (40, 70)]):
        T, ids = [], []
        n_tracks = 5 + rank
        for k in range(n_tracks):                                       # file order: track after track
            a = int(rng.integers(f0, f1 - 3)); b = int(rng.integers(a + 2, f1))
            T += [round(i / 25.0, 3) for i in range(a, b)]; ids += [k] * (b - a)
        shares.append({"T": T, "id": ids, "n_tracks": n_tracks})
        fT += T; fid += [i + base for i in ids]; base += n_tracks
    (tmp_path / "table.json").write_text(json.dumps(shares))
    script = tmp_path / "worker.py"
    script.write_text(ORDER_WORKER)
    out = str(tmp_path / "out")
    _run2(script, [ROOT, out, str(tmp_path / "table.json")], 29617, PVF_DIST_COLLECTIVE="torch")
    r0, r1 = (json.loads(open(out + ".%d" % r).read()) for r in (0, 1))
    want = formats.pandas_sort_order(fT)
    assert r0 == r1
    assert r0["T"] == [fT[i] for i in want] and r0["ids"] == [fid[i] for i in want]
    assert r0["x0"] == [round(t * 1000) for t in r0["T"]]               # the payload rows moved with their keys


def test_shard_planner_contiguous_and_balanced():
    from pyannote_video_amd import dist as pd
    shots = [(i * 250, (i + 1) * 250) for i in range(32)]
    plan = pd.shard_shots(shots, 8)
    assert plan[0][0] == 0 and plan[-1][1] == 32
    assert all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
    assert all(e - s == 4 for s, e in plan)
    uneven = [(0, 100), (100, 900), (900, 1000), (1000, 1100), (1100, 2000)]
    plan = pd.shard_shots(uneven, 2)
    assert plan == [(0, 3), (3, 5)] or plan == [(0, 2), (2, 5)]
    assert all(e > s for s, e in pd.shard_shots(shots[:3], 3))


def test_distance_shard_bounds_cover_and_balance():
    """shares of the upper triangle: contiguous, complete, and equal in AREA (pairs (a, b > a) computed) to within one track's strip"""
    from pyannote_video_amd.dist import DistanceShard
    rng = np.random.default_rng(2)
    for trial in range(50):
        T = int(rng.integers(1, 40)); world = int(rng.integers(1, 9))
        sizes = rng.integers(1, 30, T)
        rs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        cuts = DistanceShard(0, world).bounds(rs)
        assert cuts[0] == 0 and cuts[-1] == T and len(cuts) == world + 1
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
        assert [DistanceShard(r, world).track_range(rs) for r in range(world)] == list(zip(cuts, cuts[1:]))
        n = float(rs[-1])
        above = lambda a: a * n - a * a / 2.0                    # pairs (row < a, any later row)
        area = [above(float(rs[b])) - above(float(rs[a])) for a, b in zip(cuts, cuts[1:])]
        assert abs(sum(area) - n * n / 2.0) < 1e-6
        assert max(area) <= n * n / 2.0 / world + sizes.max() * n     # no share exceeds the even split by more than one track's strip
    # many equal tracks: the first rank takes few tracks (long rows of the triangle), the last one many
    rs = (np.arange(4001) * 10).astype(np.int32)
    cuts = DistanceShard(0, 8).bounds(rs)
    n_tracks = [b - a for a, b in zip(cuts, cuts[1:])]
    assert n_tracks[0] < 300 and n_tracks[-1] > 1300 and all(a <= b for a, b in zip(n_tracks, n_tracks[1:]))


def test_clip_farm_assignment_covers_every_clip_once_and_balances():
    """dist.shard_clips (configs[3]: independent clips over the ranks, no exchange step): every clip on exactly one rank; equal clips go
    round robin, clips of different lengths longest-first onto the least loaded rank"""
    from pyannote_video_amd import dist as pdist
    for n, world in ((64, 8), (64, 1), (5, 8), (0, 4), (13, 3)):
        parts = pdist.shard_clips(n, world)
        assert len(parts) == world and sorted(i for p in parts for i in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        assert all(p == sorted(p) for p in parts)
    assert pdist.shard_clips(64, 8)[3] == list(range(3, 64, 8))
    rng = np.random.default_rng(7)
    frames = rng.integers(50, 5000, 41).tolist()
    parts = pdist.shard_clips(41, 4, frames=frames)
    assert sorted(i for p in parts for i in p) == list(range(41)) and all(p == sorted(p) for p in parts)
    loads = [sum(frames[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(frames)                  # longest-processing-time-first: within one clip of each other
    assert pdist.shard_clips(41, 4, frames=frames) == parts        # deterministic: every rank computes the same table
    with pytest.raises(ValueError):
        pdist.shard_clips(3, 2, frames=[1, 2])


WORKER8 = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "pyannote-video_amd"))
import torch
import torch.distributed as dist
from pyannote_video_amd import dist as pd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
shares = json.loads(open(sys.argv[3]).read())
mine = shares[rank]
rng = np.random.default_rng(1000 + rank)
rows, n_tracks = mine["rows"], mine["tracks"]
T = (rng.random(rows) + 10 * rank) if rows else np.zeros(0)
ids = rng.integers(0, max(n_tracks, 1), rows) if rows else np.zeros(0, np.int64)
X = rng.normal(size=(rows, 128)).astype(np.float32)
gT, gid, gX, offsets = pd.gather_rows(T, ids, X, n_tracks)
gX = gX.numpy()
# the split distance step's exchange on the host form: every rank contributes its rows of a T x T table (DistanceShard.assemble)
row_start = np.array(json.loads(open(sys.argv[4]).read()), np.int32)
Tt = len(row_start) - 1
sh = pd.DistanceShard(rank, world, device="cpu")
t0, t1 = sh.track_range(row_start)
full = np.arange(Tt * Tt, dtype=np.float64).reshape(Tt, Tt)
U = sh.assemble(full[t0:t1], row_start)
before = sum(s["rows"] for s in shares[:rank])
out = {"rank": rank, "n": int(len(gT)), "offsets": offsets, "ids": gid.tolist(), "sumX": float(gX.astype(np.float64).sum()),
       "mine_back": bool(np.array_equal(gX[before:before + rows], X)), "range": [int(t0), int(t1)], "assembled": bool(np.array_equal(U, full))}
open(sys.argv[2] + ".%d" % rank, "w").write(json.dumps(out))
dist.barrier(); dist.destroy_process_group()
'''


def test_gather_rows_and_distance_shares_world8_uneven_and_empty(tmp_path):
    """The exchange step at the world size the scaling run uses, with shares a real job can produce: ranks without a single face (a
    frame range with no detection), a rank with one row, very uneven row counts.  Every rank must see the same global rows in rank order
    with its ids moved by the prefix sum of the track counts, and the split distance step's all-gather (rows cut by equal triangle area,
    some shares empty when there are fewer tracks than ranks would need) must reassemble the table on every rank."""
    import json
    from pyannote_video_amd import dist as pd
    shares = [{"rows": 11, "tracks": 3}, {"rows": 0, "tracks": 0}, {"rows": 1, "tracks": 1}, {"rows": 40, "tracks": 7},
              {"rows": 0, "tracks": 0}, {"rows": 5, "tracks": 5}, {"rows": 17, "tracks": 2}, {"rows": 3, "tracks": 1}]
    row_start = np.concatenate([[0], np.cumsum([30, 1, 2, 50, 4, 3])]).tolist()          # 6 tracks for 8 ranks: some ranks get no track
    (tmp_path / "shares.json").write_text(json.dumps(shares))
    (tmp_path / "rows.json").write_text(json.dumps(row_start))
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8)
    out = str(tmp_path / "out")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                           "--master-port", "29621", str(script), ROOT, out, str(tmp_path / "shares.json"), str(tmp_path / "rows.json")],
                          env=dict(os.environ, MASTER_ADDR="127.0.0.1", PVF_DIST_COLLECTIVE="torch", OMP_NUM_THREADS="1"), timeout=600)
    res = [json.loads(open(out + ".%d" % r).read()) for r in range(8)]
    want_off = np.concatenate([[0], np.cumsum([s["tracks"] for s in shares])[:-1]]).tolist()
    total = sum(s["rows"] for s in shares)
    for r in res:
        assert r["n"] == total and r["offsets"] == want_off
        assert r["ids"] == res[0]["ids"] and abs(r["sumX"] - res[0]["sumX"]) < 1e-9
        assert r["mine_back"] and r["assembled"]
    # the ids of rank r's rows lie in [offset_r, offset_r + tracks_r)
    o = 0
    for r, s in enumerate(shares):
        seg = res[0]["ids"][o:o + s["rows"]]
        assert all(want_off[r] <= i < want_off[r] + max(s["tracks"], 1) for i in seg)
        o += s["rows"]
    # the track ranges tile [0, T) in rank order; empty shares are allowed
    ranges = [r["range"] for r in res]
    assert ranges[0][0] == 0 and ranges[-1][1] == len(row_start) - 1 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    bounds = pd.DistanceShard(0, 8).bounds(np.array(row_start))
    assert [list(x) for x in zip(bounds, bounds[1:])] == ranges


PREFLIGHT_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "pyannote-video_amd"))
import torch.distributed as dist
from pyannote_video_amd import dist as pd
dist.init_process_group("gloo")
lines = []
rep = pd.preflight(log=lines.append, payload_bytes=1 << 16)
open(sys.argv[2] + ".%d" % dist.get_rank(), "w").write(json.dumps({"rep": rep, "lines": lines}))
dist.barrier(); dist.destroy_process_group()
'''


def test_preflight_world2(tmp_path):
    """dist.preflight (what `bench.py --gpus N` runs before it renders a frame): communicator, counts, an all-gather in uneven shares with
    every byte verified, each step reported per rank -- here over gloo at world 2 (PVF_DIST_COLLECTIVE=torch)"""
    import json
    script = tmp_path / "preflight.py"
    script.write_text(PREFLIGHT_WORKER)
    out = str(tmp_path / "out")
    _run2(script, [ROOT, out], 29641, PVF_DIST_COLLECTIVE="torch")
    for r in (0, 1):
        d = json.loads(open(out + ".%d" % r).read())
        rep = d["rep"]
        assert rep["world"] == 2 and rep["rank"] == r and rep["collective"] == "torch"
        part = (1 << 16) // 3 // 16 * 16
        assert rep["allgatherv_bytes"] == part + 2 * part
        assert len(d["lines"]) == 3 and all(("rank %d/2" % r) in l for l in d["lines"]) and "verified" in d["lines"][-1]
