"""f4 -- shot boundary detection (reference: pyannote/video/structure/shot.py).

CPU: the oracle's Farneback restatement behaves like an optical flow; the reference's own Shot class, executed verbatim on a `cv2` made
of the oracle, gives the segments the product's host logic gives (median filter, threshold, run suppression: shot.py:119-147).
GPU: gray images, flows and displaced frame differences of csrc/shot.hip equal the oracle bit for bit; the product's Shot finds the cuts
of a synthetic clip.  OpenCV itself is not available: its arithmetic is restated, PARITY UNPINNED (oracle/pvo_shot.c)."""
import os
import sys
import types

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import refhost  # noqa: E402


def _textured(seed, h=140, w=120):
    import scipy.ndimage as ndi
    base = ndi.gaussian_filter(np.random.default_rng(seed).uniform(0, 255, (h, w)), 3)
    return (base - base.min()) / (base.max() - base.min()) * 255


def test_tables_of_product_and_oracle_agree():
    from oracle import oracle
    from pyannote_video_amd import structure
    a, b = structure.shot_tables(), oracle.shot_tables()
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert abs(float(a[0] + 2 * a[1:6].sum()) - 1.0) < 1e-6          # normalised Gaussian


@pytest.mark.parametrize("shift", [(1, 2), (-2, 1), (0, 3)])
def test_oracle_flow_recovers_a_translation(shift):
    """content moved by (sx, sy) between the two images => flow ~ (-sx, -sy) ... in OpenCV's sense prev(x, y) ~ cur(x + fx, y + fy)"""
    from oracle import oracle
    sx, sy = shift
    base = _textured(3)
    prev = base[20:108, 20:70].astype(np.uint8)
    cur = base[20 + sy:108 + sy, 20 + sx:70 + sx].astype(np.uint8)      # cur(x, y) = prev(x + sx, y + sy)
    flow = oracle.farneback_small(prev, cur)
    centre = flow[25:60, 15:35].reshape(-1, 2).mean(0)
    assert abs(centre[0] + sx) < 0.25 and abs(centre[1] + sy) < 0.25, centre
    # (the reference's dfd adds the flow's x component to y and its y component to x -- `dy, dx = flow[y, x]`, shot.py:91 -- so a displaced
    # lookup only "explains" motion along the diagonal; that is kept as it is.  Unrelated content scores far above an unchanged frame.)
    same = oracle.shot_dfd(prev, prev)
    other = oracle.shot_dfd(prev, _textured(9)[20:108, 20:70].astype(np.uint8))
    assert same < 4.0 and other > 5 * same


def test_pyramid_levels_follow_opencvs_rule():
    """[EXT optflowgf.cpp]: halve while both sides stay >= 32, at most three times"""
    from oracle import oracle
    assert [oracle.farneback_levels(h, w) for h, w in ((88, 50), (64, 63), (64, 64), (177, 100), (128, 300), (284, 160), (256, 256), (1000, 1000))] == \
        [0, 0, 1, 1, 2, 2, 3, 3]


@pytest.mark.parametrize("size", [(177, 100), (284, 160), (330, 280)])       # one, two and three coarser levels
def test_oracle_flow_on_coarser_levels_recovers_a_translation(size):
    """Shot(height >= 64) (scripts/pyannote-structure.py:45,111): OpenCV's Farneback starts on coarser pyramid levels; a shift larger than
    the single-level case can follow (5 px) is recovered, and the displaced difference of a moved frame stays far below a changed one"""
    from oracle import oracle
    h, w = size
    assert oracle.farneback_levels(h, w) == {100: 1, 160: 2, 280: 3}[w]
    base = _textured(4, h + 40, w + 40)
    prev = base[20:20 + h, 20:20 + w].astype(np.uint8)
    cur = base[20 + 5:20 + 5 + h, 20 + 5:20 + 5 + w].astype(np.uint8)      # cur(x, y) = prev(x + 5, y + 5): diagonal, so the reference's
    flow = oracle.farneback(prev, cur)                                     # swapped lookup (`dy, dx = flow[y, x]`) explains it too
    centre = flow[h // 4:3 * h // 4, w // 4:3 * w // 4].reshape(-1, 2)
    assert np.abs(np.median(centre, 0) + 5).max() < 0.3, np.median(centre, 0)
    moved = oracle.shot_dfd(prev, cur)
    other = oracle.shot_dfd(prev, _textured(9, h + 40, w + 40)[20:20 + h, 20:20 + w].astype(np.uint8))
    assert other > 3 * moved


class _Clip(object):
    """the slice of the reference's Video that Shot touches (shot.py:55-69,101-117,119-147)"""

    def __init__(self, frames, fps=25.0):
        self.frames, self.frame_rate = frames, fps
        self._size = (frames[0].shape[1], frames[0].shape[0])
        self.step = 1.0 / fps
        self.start, self.end = 0.0, len(frames) / fps

    def __iter__(self):
        for i, f in enumerate(self.frames):
            yield i / self.frame_rate, f


def _cut_clip(n=36, cuts=(11, 24), h=120, w=160, seed=5):
    """gently moving textured backgrounds with hard cuts at the given frame indices"""
    rng = np.random.default_rng(seed)
    frames, k = [], 0
    bg = _textured(seed, h + 40, w + 40)
    for i in range(n):
        if i in cuts:
            k += 1
            bg = _textured(seed + 10 * k, h + 40, w + 40)
        o = 10 + (i % 7)
        g = bg[o:o + h, o // 2:o // 2 + w]
        rgb = np.stack([g, np.clip(g * 0.9 + 10, 0, 255), np.clip(g * 1.05, 0, 255)], axis=2)
        frames.append(np.clip(rgb + rng.normal(0, 1.5, rgb.shape), 0, 255).astype(np.uint8))
    return frames


def _oracle_cv2(oracle):
    cv2 = types.ModuleType("cv2")
    cv2.__version__ = "4.2.0"
    cv2.COLOR_RGB2GRAY = 7

    class _Deferred(object):                    # cvtColor's result is only ever handed to resize (shot.py:71-73)
        def __init__(self, rgb):
            self.rgb = rgb

    def cvtColor(rgb, code):
        assert code == cv2.COLOR_RGB2GRAY
        return _Deferred(rgb)

    def resize(img, dsize):
        return oracle.shot_convert(img.rgb, dsize[0], dsize[1])

    def calcOpticalFlowFarneback(prev, cur, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
        assert (flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags) == (None, 0.5, 3, 15, 3, 5, 1.1, 0)
        return oracle.farneback(prev, cur)
    cv2.cvtColor, cv2.resize, cv2.calcOpticalFlowFarneback = cvtColor, resize, calcOpticalFlowFarneback
    return cv2


@pytest.mark.skipif(not refhost.have_reference(), reason="/root/reference is not present (GPU box)")
@pytest.mark.parametrize("height", [50, 100])                     # the reference's default (one level) and a two-level size
def test_reference_shot_class_verbatim_equals_product_host_logic(height):
    from oracle import oracle
    from pyannote_video_amd import structure
    from pyannote_video_amd._core import Segment
    frames = _cut_clip()
    clip = _Clip(frames)
    with refhost.reference_shot_module(_oracle_cv2(oracle), Segment) as ref:
        shot = ref.Shot(clip, height=height, context=0.4, threshold=1.0)
        ref_pairs = list(shot.iter_dfd())
        ref_segments = [(s.start, s.end) for s in shot]
        ksize = shot._kernel_size
        assert shot._resize == (height, int(160 * height / 120))
    # the oracle's displaced frame difference == what the reference's per-pixel Python loop computes from the same flow
    ow, oh = height, int(160 * height / 120)
    small = [oracle.shot_convert(f, ow, oh) for f in frames]
    mine = [(i / 25.0, oracle.shot_dfd(small[i - 1], small[i])) for i in range(1, len(frames))]
    assert [t for t, _ in mine] == [t for t, _ in ref_pairs]
    assert [d for _, d in mine] == [float(d) for _, d in ref_pairs]
    # and the product's thresholding == the reference's __iter__
    got = [(s.start, s.end) for s in structure.boundaries([t for t, _ in mine], [d for _, d in mine], clip.start, clip.end, ksize, 1.0)]
    assert got == ref_segments
    found = [round(b * 25) for _, b in got[:-1]]
    assert found[:2] == [11, 24] and all(f >= 34 for f in found[2:])   # the two cuts (+ the median filter's zero-padded last frames, as in the reference)


# ---------------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_shot_dfd_bit_exact_and_cuts_found(ctx):
    from oracle import oracle
    from pyannote_video_amd import structure
    frames = _cut_clip(n=30, cuts=(9, 20))
    tables = structure.shot_tables()
    ow, oh = 50, int(160 * 50 / 120)
    dfd, gray, flow = ctx.shot_dfd(frames, ow, oh, tables, want_gray=True, want_flow=True)
    small = [oracle.shot_convert(f, ow, oh) for f in frames]
    assert np.array_equal(gray, np.stack(small))
    for i in range(1, len(frames)):
        f = oracle.farneback_small(small[i - 1], small[i], tables)
        assert np.array_equal(flow[i - 1], f), (i, np.abs(flow[i - 1] - f).max())
        assert dfd[i - 1] == oracle.shot_dfd(small[i - 1], small[i], tables)
    clip = _Clip(frames)
    shot = structure.Shot(clip, height=50, context=0.4, threshold=1.0, ctx=ctx, chunk=8)      # several overlapping chunks
    assert [d for _, d in shot.iter_dfd()] == dfd.tolist()
    segments = [(s.start, s.end) for s in shot]
    found = [round(b * 25) for _, b in segments[:-1]]
    assert found[:2] == [9, 20] and all(f >= 28 for f in found[2:])
    assert segments[0][0] == 0.0 and segments[-1][1] == clip.end


@pytest.mark.gpu
@pytest.mark.parametrize("height", [100, 160, 300])              # one, two and three coarser pyramid levels (Shot(height=...), shot.py:53-60)
def test_gpu_shot_dfd_coarser_levels_bit_exact(ctx, height):
    from oracle import oracle
    from pyannote_video_amd import structure
    frames = _cut_clip(n=8, cuts=(4,), h=360, w=480)
    tables = structure.shot_tables()
    ow, oh = height, int(480 * height / 360)
    assert oracle.farneback_levels(oh, ow) == {100: 1, 160: 2, 300: 3}[height]
    dfd, gray, flow = ctx.shot_dfd(frames, ow, oh, tables, want_gray=True, want_flow=True)
    small = [oracle.shot_convert(f, ow, oh) for f in frames]
    assert np.array_equal(gray, np.stack(small))
    for i in range(1, len(frames)):
        f = oracle.farneback(small[i - 1], small[i], tables)
        assert np.array_equal(flow[i - 1], f), (i, np.abs(flow[i - 1] - f).max())
        assert dfd[i - 1] == oracle.shot_dfd(small[i - 1], small[i], tables)
    assert int(np.argmax(dfd)) == 3                                # the cut at frame 4 (the clip's background also jumps back at frame 7)
    shot = structure.Shot(_Clip(frames), height=height, context=0.2, threshold=1.0, ctx=ctx)
    assert [d for _, d in shot.iter_dfd()] == dfd.tolist()


@pytest.mark.gpu
def test_gpu_shot_dfd_1080p_pair(ctx):
    from oracle import oracle
    from pyannote_video_amd import structure, synth
    video = synth.SyntheticVideo(width=1920, height=1080, n_frames=4, n_shots=2, faces=3, seed=11)
    frames = [video.frame(i) for i in range(4)]
    tables = structure.shot_tables()
    ow, oh = 50, int(1920 * 50 / 1080)
    dfd, gray = ctx.shot_dfd(frames, ow, oh, tables, want_gray=True)
    small = [oracle.shot_convert(f, ow, oh) for f in frames]
    assert np.array_equal(gray, np.stack(small))
    assert dfd.tolist() == [oracle.shot_dfd(small[i], small[i + 1], tables) for i in range(3)]
    assert dfd[1] > 2 * max(dfd[0], dfd[2])                          # the shot cut between frames 1 and 2


@pytest.mark.gpu
def test_cli_shot_verb_writes_a_shot_file_track_reads(tmp_path):
    """`shot` writes the Timeline file `track` takes as <shot.json> (scripts/pyannote-structure.py:65-70 -> pyannote-face.py:253-254)"""
    import json
    from pyannote_video_amd import cli, synth
    out = str(tmp_path / "shots.json")
    assert cli.main(["--fps", "25", "shot", "synthetic:640x360x40:2:3:7", out, "--window", "0.4"]) == 0
    data = json.load(open(out))
    assert data["pyannote"] == "Timeline" and len(data["content"]) >= 2
    shots = cli.load_shots(out)
    video = synth.SyntheticVideo(width=640, height=360, n_frames=40, n_shots=2, faces=3, seed=7)
    cut = video.shot_bounds[1] / 25.0
    assert any(abs(s.end - cut) < 1e-9 for s in shots[:-1])            # the clip's own shot cut is among the boundaries
    assert shots[0].start == 0.0 and abs(shots[-1].end - video.duration) < 1e-9
