"""Shot boundary detection: the reference's `Shot` (pyannote/video/structure/shot.py) with the displaced
frame differences computed on the GPU -- the producer of the shot file that `track` reads and that multi-GPU jobs are cut along.

The reference converts every frame to a 50-pixel-wide gray image, runs OpenCV's Farneback optical flow between consecutive images, and
walks every pixel in Python to build the displaced frame (shot.py:75-99).  Here the frames already staged in HBM are reduced to the small
gray images in one launch and every consecutive pair is one workgroup (csrc/shot.hip).  What happens to the differences afterwards --
median filtering, the threshold on the normalised difference, "first of a run of consecutive frames" -- is the reference's code path
(shot.py:119-147), kept in Python.  OpenCV's arithmetic is restated, not linked: PARITY UNPINNED (see oracle/pvo_shot.c).
"""
import numpy as np

try:                                    # the reference's own segment type where it is installed (structure/shot.py:33) ...
    from pyannote.core import Segment
except ImportError:                     # ... a stand-in with the same constructor and truthiness where it is not
    from ._core import Segment


def shot_tables(poly_n=5, poly_sigma=1.1):
    """the 22 floats both the kernels and the oracle work from: the normalised Gaussian g[x], x g[x], x^2 g[x] for x = 0..5 and the four
    entries (1,1), (0,3), (3,3), (5,5) of the inverse moment matrix of the polynomial expansion (Farneback 2003; computed in double)"""
    if poly_n != 5:
        raise ValueError("the kernels are written for poly_n = 5 (the reference's value, shot.py:80-84)")
    n = poly_n
    x = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-x * x / (2.0 * poly_sigma * poly_sigma))
    g = (g * (1.0 / g.sum())).astype(np.float32)
    t = np.zeros(22, np.float32)
    k = np.arange(0, n + 1)
    t[0:6] = g[n:]
    t[6:12] = (k * g[n:]).astype(np.float32)
    t[12:18] = (k * k * g[n:]).astype(np.float32)
    gd = g.astype(np.float64)
    w = np.outer(gd, gd)                               # w[y, x]
    X, Y = np.meshgrid(x, x)
    G = np.zeros((6, 6))
    G[0, 0] = w.sum(); G[1, 1] = (w * X * X).sum(); G[3, 3] = (w * X ** 4).sum(); G[5, 5] = (w * X * X * Y * Y).sum()
    G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = G[1, 1]
    G[4, 4] = G[3, 3]
    G[3, 4] = G[4, 3] = G[5, 5]
    inv = np.linalg.inv(G)
    t[18], t[19], t[20], t[21] = inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]
    return t


def boundaries(times, dfd, start, end, kernel_size, threshold):
    """The reference's decision rule (shot.py:119-147) on the displaced frame differences dfd[i] taken at times[i]: a frame is a
    candidate when its difference exceeds the median-filtered one by more than `threshold` times that median; of a run of candidates at
    consecutive indices only the first counts -- where "consecutive" is judged against the previous candidate, or against index 0 for the
    first one (so a candidate at index 1 never counts: the reference's loop starts from `_i = 0`).  Segments run from cut to cut; the
    last one is kept if it is not empty.  Array form of that loop; tests/test_shot.py holds it equal to the reference class run verbatim."""
    import scipy.signal
    y = np.asarray(dfd, np.float64)
    base = scipy.signal.medfilt(y, kernel_size=kernel_size)
    with np.errstate(divide="ignore", invalid="ignore"):
        excess = (y - base) / base
    candidates = np.flatnonzero(excess > threshold)
    before = np.concatenate(([0], candidates[:-1]))
    cuts = candidates[candidates != before + 1]
    edges = [start] + [times[int(i)] for i in cuts] + [end]
    segments = [Segment(a, b) for a, b in zip(edges[:-1], edges[1:])]
    return segments[:-1] + ([segments[-1]] if segments[-1] else [])


class Shot(object):
    """Shot boundary detection based on displaced frame difference (shot.py:40-69)

    Parameters
    ----------
    video : iterable of (t, rgb) with `_size` (width, height), `step`, `start`, `end` like the reference's Video
    height : int, optional      the small image is this many pixels WIDE (the reference hands (height, int(w * height / h)) to cv2.resize as
                                (width, height)).  Defaults to 50 (one pyramid level of the optical flow; a side of 64 pixels or more brings
                                OpenCV's coarser levels, computed in the same kernel).
    context : float, optional   median filtering context in seconds.  Defaults to 2.
    threshold : float, optional Defaults to 1.
    ctx : runtime.Context
    chunk : frames reduced and compared per call (consecutive chunks overlap by one frame)
    """

    def __init__(self, video, height=50, context=2.0, threshold=1.0, ctx=None, chunk=1024):
        self.video = video
        self.height = height
        self.threshold = threshold
        self.context = context
        frame_w, frame_h = self.video._size
        # (shot.py:62) handed to cv2.resize as dsize, i.e. (width, height) of the small image
        self._resize = (self.height, int(frame_w * self.height / frame_h))
        # (shot.py:65-67) median window in frames: odd, at least 3
        self._kernel_size = max(3, int(np.ceil(self.context / self.video.step) // 2 * 2 + 1))
        if ctx is None:
            from .runtime import Context
            ctx = Context(0)
        self.ctx = ctx
        self.chunk = int(chunk)
        self._tables = shot_tables()

    def iter_dfd(self):
        """Pairwise displaced frame difference: (t of the later frame, dfd), like shot.py:101-117"""
        ow, oh = self._resize
        pending_t, pending_f = [], []
        for t, rgb in self.video:
            pending_t.append(t)
            pending_f.append(rgb)
            if len(pending_f) == self.chunk:
                for item in self._flush(pending_t, pending_f, ow, oh):
                    yield item
                pending_t, pending_f = pending_t[-1:], pending_f[-1:]       # the next chunk starts with this chunk's last frame
        if len(pending_f) > 1:
            for item in self._flush(pending_t, pending_f, ow, oh):
                yield item

    def _flush(self, ts, frames, ow, oh):
        dfd = self.ctx.shot_dfd(frames, ow, oh, self._tables)
        return list(zip(ts[1:], dfd.tolist()))

    def __iter__(self):
        pairs = list(self.iter_dfd())
        if not pairs:
            last = Segment(self.video.start, self.video.end)
            if last:
                yield last
            return
        t, y = zip(*pairs)
        for segment in boundaries(t, y, self.video.start, self.video.end, self._kernel_size, self.threshold):
            yield segment
