"""Multi-GPU: one process per GPU; a long video is cut into contiguous frame ranges at shot boundaries, every rank runs
detect -> track -> extract on its range with no data-path collective, then ONE exchange step: an all-gather of the
per-rank (time, track id, 128-D embedding) rows over RCCL/xGMI, followed by a single global clustering.

Track ids: the reference numbers tracks in yield order over the whole video (pyannote-face.py:261) and tracks never span
shots (tracking.py:359-362,410-417), so global id = local id + exclusive prefix sum of the per-rank track counts."""
import numpy as np


def shard_shots(shot_ranges, world_size):
    """Contiguous groups of shots per rank, balancing frame counts greedily. shot_ranges: [(i0, i1)]. -> [(s0, s1)] per rank"""
    n = len(shot_ranges)
    total = sum(b - a for a, b in shot_ranges)
    out, s = [], 0
    acc = 0
    for r in range(world_size):
        if r == world_size - 1:
            out.append((s, n))
            break
        target = total * (r + 1) / float(world_size)
        e = s
        while e < n - (world_size - 1 - r) and (acc + (shot_ranges[e][1] - shot_ranges[e][0]) <= target or e == s):
            acc += shot_ranges[e][1] - shot_ranges[e][0]
            e += 1
        out.append((s, e))
        s = e
    return out


def _gather_padded(loc, dev):
    """all-gather of float64 [n_r, k] blocks of different n_r: one count exchange + one padded payload exchange"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    cnt = torch.tensor([len(loc)], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(allc, cnt)
    rows = [int(c[0]) for c in allc]
    pay = torch.zeros((max(max(rows), 1), loc.shape[1]), dtype=torch.float64, device=dev)
    if len(loc):
        pay[:len(loc)] = torch.from_numpy(np.ascontiguousarray(loc, np.float64)).to(dev)
    allp = [torch.zeros_like(pay) for _ in range(world)]
    dist.all_gather(allp, pay)
    return [allp[r][:rows[r]].cpu().numpy() for r in range(world)]


def gather_rows(face_T, face_id, X, n_tracks, device=None, file_T=None, file_id=None):
    """All-gather variable-length rows from every rank.  Returns (T, id_global, X) concatenated in rank order and the
    per-rank track offsets.  Uses torch.distributed (backend nccl == RCCL on ROCm, gloo on CPU).

    file_T / file_id: this rank's share of the track table in file order (FacePipeline.run(..., reorder=False)["file_T"/"file_id"]).
    When given, the shares are gathered too (a few bytes per row) and the rows are put into the order the reference's `extract`
    writes them for the WHOLE video (formats.file_order: pandas' unstable sort of the complete table decides the order of the
    faces of one frame), so a sharded run returns exactly the rows of a single-process run."""
    import torch
    import torch.distributed as dist
    from . import formats
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        T, ids, Xa = np.asarray(face_T, np.float64), np.asarray(face_id, np.int64), np.asarray(X, np.float64)
        if file_T is not None and len(T):
            perm = formats.file_order(T, ids, file_T, file_id)
            T, ids, Xa = T[perm], ids[perm], Xa[perm]
        return T, ids, Xa, [0]
    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    counts = torch.tensor([len(face_T), int(n_tracks)], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(allc, counts)
    rows = [int(c[0]) for c in allc]
    tracks = [int(c[1]) for c in allc]
    offsets = [0]
    for k in tracks[:-1]:
        offsets.append(offsets[-1] + k)
    cap = max(max(rows), 1)
    # one padded payload per rank: [cap, 130] float64 = (T, local id, 128 values); <= 32 MB/rank even at 3e4 rows
    pay = torch.zeros((cap, 130), dtype=torch.float64, device=dev)
    if len(face_T):
        loc = np.concatenate([np.asarray(face_T, np.float64)[:, None], np.asarray(face_id, np.float64)[:, None],
                              np.asarray(X, np.float64)], axis=1)
        pay[:len(face_T)] = torch.from_numpy(loc).to(dev)
    allp = [torch.zeros_like(pay) for _ in range(world)]
    dist.all_gather(allp, pay)
    Ts, ids, Xs = [], [], []
    for r in range(world):
        a = allp[r][:rows[r]].cpu().numpy()
        Ts.append(a[:, 0]); ids.append(a[:, 1].astype(np.int64) + offsets[r]); Xs.append(a[:, 2:])
    T, gid, Xa = np.concatenate(Ts), np.concatenate(ids), np.ascontiguousarray(np.concatenate(Xs))
    if file_T is not None:
        parts = _gather_padded(np.stack([np.asarray(file_T, np.float64), np.asarray(file_id, np.float64)], 1).reshape(-1, 2), dev)
        fT = np.concatenate([p[:, 0] for p in parts])
        fid = np.concatenate([p[:, 1].astype(np.int64) + offsets[r] for r, p in enumerate(parts)])
        if len(T):
            perm = formats.file_order(T, gid, fT, fid)
            T, gid, Xa = T[perm], gid[perm], np.ascontiguousarray(Xa[perm])
    return T, gid, Xa, offsets


class DistanceShard(object):
    """Splits the N x N pairwise distances of the global clustering over the ranks: every rank already holds all gathered rows,
    computes the rows of the T x T track-pair matrix for a contiguous share of the tracks (balanced by row count, so by work),
    and one all-gather of those rows (T x T doubles in total) gives every rank the complete matrix.  Without it each rank would
    repeat the whole O(N^2) step, which grows with the square of the number of GPUs under weak scaling."""

    def __init__(self, rank, world, device=None):
        self.rank, self.world, self.device = rank, world, device

    def bounds(self, row_start):
        n = int(row_start[-1])
        T = len(row_start) - 1
        cuts = [0]
        for r in range(1, self.world):
            target = n * r / float(self.world)
            t = cuts[-1]
            while t < T and row_start[t + 1] <= target:
                t += 1
            cuts.append(t)
        cuts.append(T)
        return cuts

    def track_range(self, row_start):
        cuts = self.bounds(row_start)
        return cuts[self.rank], cuts[self.rank + 1]

    def assemble(self, D_mine, row_start):
        import torch
        import torch.distributed as dist
        T = D_mine.shape[0]
        cuts = self.bounds(row_start)
        dev = self.device if self.device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
        t = torch.from_numpy(np.ascontiguousarray(D_mine)).to(dev)
        parts = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t)
        D = np.zeros((T, T), np.float64)
        for r in range(self.world):
            a, b = cuts[r], cuts[r + 1]
            if b > a:
                D[a:b] = parts[r][a:b].cpu().numpy()
        return D


def global_cluster(clustering, face_T, face_id, X):
    """single global clustering on the gathered rows (computed identically on every rank)"""
    if len(face_T) == 0:
        return {}
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and clustering.shard is None:
        clustering.shard = DistanceShard(dist.get_rank(), dist.get_world_size())
    sp, data = clustering.model.preprocess((face_T, face_id, X))
    res = clustering(sp, features=data)
    return {int(track): int(label) for _, track, label in res.itertracks(yield_label=True)}
