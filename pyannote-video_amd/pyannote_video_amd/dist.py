"""Multi-GPU: one process per GPU; a long video is cut into contiguous frame ranges at shot boundaries, every rank runs
detect -> track -> extract on its range with no data-path collective, then ONE exchange step: an all-gather of the
per-rank (time, track id, 128-D embedding) rows over RCCL/xGMI, followed by a single global clustering.

Track ids: the reference numbers tracks in yield order over the whole video (pyannote-face.py:261) and tracks never span
shots (tracking.py:359-362,410-417), so global id = local id + exclusive prefix sum of the per-rank track counts."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DIST_LIB_PATH = os.path.join(_HERE, "libpvface_dist.so")
_DIST_SIGS = {
    "pvfd_last_error": (C.c_char_p, []),
    "pvfd_unique_id": (C.c_int32, [C.c_void_p]),
    "pvfd_comm_create": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pvfd_comm_destroy": (C.c_int32, [C.c_uint64]),
    "pvfd_allgather_rows": (C.c_int32, [C.c_uint64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pvfd_max_rows": (C.c_int32, [C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]),
}
DIST_EXPORTS = sorted(_DIST_SIGS)
_dist_lib = None


def dist_lib():
    """libpvface_dist.so (include/pvface_dist.h): the RCCL all-gather behind a C ABI"""
    global _dist_lib
    if _dist_lib is None:
        l = C.CDLL(DIST_LIB_PATH)
        for name, (res, args) in _DIST_SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _dist_lib = l
    return _dist_lib


class RcclRows(object):
    """one communicator of libpvface_dist.so per process: all-gather of float64 rows with a different count per rank"""

    def __init__(self, device, rank, world, unique_id):
        self.l = dist_lib()
        self.rank, self.world = rank, world
        h = C.c_uint64(0)
        idb = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._check(self.l.pvfd_comm_create(int(device), int(rank), int(world), idb, C.byref(h)))
        self.h = h.value

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("libpvface_dist: " + self.l.pvfd_last_error().decode("utf-8", "replace"))

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        l = dist_lib()
        if l.pvfd_unique_id(buf) != 0:
            raise RuntimeError("libpvface_dist: " + l.pvfd_last_error().decode("utf-8", "replace"))
        return bytes(buf)

    def allgather_rows(self, rows):
        """rows float64 [n, k] (n may differ per rank, k must not) -> (all rows in rank order [N, k], counts per rank)"""
        rows = np.ascontiguousarray(rows, np.float64)
        n, k = rows.shape
        counts = np.zeros(self.world, np.int64)
        total = C.c_int64(0)
        self._check(self.l.pvfd_max_rows(self.h, n, counts.ctypes.data_as(C.c_void_p), C.byref(total)))
        out = np.zeros((max(total.value, 1), k), np.float64)
        self._check(self.l.pvfd_allgather_rows(self.h, rows.ctypes.data_as(C.c_void_p), n, k, counts.ctypes.data_as(C.c_void_p),
                                               out.ctypes.data_as(C.c_void_p), len(out), C.byref(total)))
        return out[:total.value], [int(c) for c in counts]

    def close(self):
        if getattr(self, "h", None):
            self.l.pvfd_comm_destroy(self.h)
            self.h = None


_rccl = {"tried": False, "comm": None}


def rccl_rows():
    """The process's RCCL communicator of libpvface_dist.so, or None when the job does not run on GPUs over nccl (CPU tests over gloo),
    when PVF_DIST_COLLECTIVE=torch asks for the torch.distributed collectives, or when any rank failed to set it up (all ranks then
    agree to use torch.distributed instead -- decided with one all-reduce, so no rank is left waiting in a collective)."""
    import torch
    import torch.distributed as dist
    if _rccl["tried"]:
        return _rccl["comm"]
    _rccl["tried"] = True
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    if dist.get_backend() != "nccl" or os.environ.get("PVF_DIST_COLLECTIVE", "rccl") == "torch":
        return None
    rank, world = dist.get_rank(), dist.get_world_size()
    comm, ok = None, 1
    try:
        box = [RcclRows.unique_id() if rank == 0 else None]
    except Exception:
        box, ok = [None], 0
    dist.broadcast_object_list(box, src=0)
    try:
        if box[0] is None:
            raise RuntimeError("no communicator id")
        comm = RcclRows(torch.cuda.current_device(), rank, world, box[0])
    except Exception as e:                      # noqa: BLE001 -- any failure means: use the torch path, together
        import sys
        sys.stderr.write("[pvface] libpvface_dist unavailable on rank %d (%s); using torch.distributed collectives\n" % (rank, e))
        ok, err = 0, e
    flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        if comm is not None:
            comm.close()
        comm = None
        if os.environ.get("PVF_DIST_STRICT", "0") == "1":
            # a job that asked for the C-ABI collective (bench.py --gpus N > 1 does) must not measure something else in its place
            raise RuntimeError("libpvface_dist.so: the RCCL communicator could not be set up on every rank and PVF_DIST_STRICT=1 forbids "
                               "the torch.distributed fallback (set PVF_DIST_COLLECTIVE=torch to ask for it explicitly)")
    _rccl["comm"] = comm
    return comm


def collective_name():
    """which exchange step a multi-GPU run uses: 'libpvface_dist' (RCCL behind the C ABI), 'torch' (torch.distributed collectives) or
    'none' (a single process)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return "none"
    return "libpvface_dist" if rccl_rows() is not None else "torch"


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


def shard_clips(n_clips, world_size, frames=None):
    """The clip farm (BASELINE.json configs[3]: independent videos, one result each, no exchange step): which clips every rank takes.
    Without `frames`: round robin (clips of one length).  With `frames` = frame count per clip: longest first onto the rank with the least
    work so far (ties: lowest rank), each rank's clips then in input order.  -> [[clip indices] per rank]; every clip exactly once."""
    out = [[] for _ in range(world_size)]
    if frames is None:
        for i in range(n_clips):
            out[i % world_size].append(i)
        return out
    if len(frames) != n_clips:
        raise ValueError("shard_clips: one frame count per clip")
    load = [0] * world_size
    for i in sorted(range(n_clips), key=lambda k: (-int(frames[k]), k)):
        r = min(range(world_size), key=lambda q: (load[q], q))
        out[r].append(i)
        load[r] += int(frames[i])
    return [sorted(c) for c in out]


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
    rc = rccl_rows()
    if rc is not None:
        # the C-ABI collective (libpvface_dist.so): one row of (T, local id, 128 values, this rank's track count) per face; a rank
        # without faces still announces its track count with a marker row
        nT = len(face_T)
        loc = np.zeros((max(nT, 1), 131), np.float64)
        loc[:, 130] = float(n_tracks)
        if nT:
            loc[:, 0] = np.asarray(face_T, np.float64); loc[:, 1] = np.asarray(face_id, np.float64); loc[:, 2:130] = np.asarray(X, np.float64)
        else:
            loc[0, 1] = -1.0
        allrows, counts = rc.allgather_rows(loc)
        tracks, o = [], 0
        for cnt in counts:
            tracks.append(int(allrows[o, 130])); o += cnt
        offsets = [0]
        for k in tracks[:-1]:
            offsets.append(offsets[-1] + k)
        Ts, ids, Xs, o = [], [], [], 0
        for r, cnt in enumerate(counts):
            a = allrows[o:o + cnt]; o += cnt
            a = a[a[:, 1] >= 0]
            Ts.append(a[:, 0]); ids.append(a[:, 1].astype(np.int64) + offsets[r]); Xs.append(a[:, 2:130])
        T, gid, Xa = np.concatenate(Ts), np.concatenate(ids), np.ascontiguousarray(np.concatenate(Xs))
        if file_T is not None:
            ft = np.stack([np.asarray(file_T, np.float64), np.asarray(file_id, np.float64)], 1).reshape(-1, 2)
            marker = len(ft) == 0
            fall, fcounts = rc.allgather_rows(ft if not marker else np.array([[0.0, -1.0]]))
            fT, fid, o = [], [], 0
            for r, cnt in enumerate(fcounts):
                a = fall[o:o + cnt]; o += cnt
                a = a[a[:, 1] >= 0]
                fT.append(a[:, 0]); fid.append(a[:, 1].astype(np.int64) + offsets[r])
            if len(T):
                perm = formats.file_order(T, gid, np.concatenate(fT), np.concatenate(fid))
                T, gid, Xa = T[perm], gid[perm], np.ascontiguousarray(Xa[perm])
        return T, gid, Xa, offsets
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
        rc = rccl_rows()
        if rc is not None:
            a, b = cuts[self.rank], cuts[self.rank + 1]
            allrows, counts = rc.allgather_rows(np.ascontiguousarray(D_mine[a:b]).reshape(b - a, T))
            assert counts == [cuts[r + 1] - cuts[r] for r in range(self.world)] and len(allrows) == T
            return np.ascontiguousarray(allrows)
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
