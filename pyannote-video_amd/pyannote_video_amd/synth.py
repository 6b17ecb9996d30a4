"""Deterministic synthetic video for the BASELINE.json configs (no sample video / ffmpeg exists here).

`SyntheticVideo` exposes the slice of the reference `Video` interface the face path consumes
(reference video.py:96-187, 411-464): iteration yields `(t, frame)` with `frame` a C-contiguous
`uint8[H, W, 3]` RGB array and `t = i / frame_rate` (np.arange semantics of video.py:432), plus
`frame_rate`, `size`, `frame_size`, `duration`.

Content: per-shot low-pass background + sensor-noise tile, `faces` procedurally rendered
face-like patches per frame (skin ellipse, hair, brows, eyes, nose, mouth) following smooth trajectories inside
disjoint regions, drawn from `identities` appearance parameter sets so clustering has ground truth.
"""
import numpy as np

POSES = ("frontal", "yaw_left", "yaw_right", "roll_left", "roll_right")


def identity_params(k):
    """Appearance parameters of identity k (deterministic)."""
    rng = np.random.default_rng(7919 + 31 * k)
    skin = np.array([rng.uniform(150, 235), rng.uniform(110, 190), rng.uniform(80, 170)])
    return {
        "skin": skin,
        "hair": np.array([rng.uniform(10, 120), rng.uniform(10, 90), rng.uniform(10, 70)]),
        "hair_frac": rng.uniform(0.14, 0.30),
        "eye_dx": rng.uniform(0.16, 0.20),
        "eye_r": (rng.uniform(0.065, 0.095), rng.uniform(0.030, 0.050)),
        "mouth_r": (rng.uniform(0.12, 0.20), rng.uniform(0.035, 0.075)),
        "mouth_col": np.array([rng.uniform(90, 190), rng.uniform(20, 70), rng.uniform(30, 80)]),
        "mark_col": np.array([rng.uniform(0, 255), rng.uniform(0, 255), rng.uniform(0, 255)]),
        "mark_pos": (rng.uniform(0.25, 0.75), rng.uniform(0.50, 0.62)),
        "mark_r": rng.uniform(0.05, 0.09),
        "cheek": np.array([rng.uniform(160, 255), rng.uniform(60, 160), rng.uniform(60, 160)]),
    }


def _cover(q, r_px):
    """anti-aliased coverage of the ellipse q <= 1 (q = normalised squared radius), r_px = smaller radius in px"""
    d = (np.sqrt(np.maximum(q, 1e-12)) - 1.0) * r_px
    return np.clip(0.5 - d, 0.0, 1.0)


def render_face(size, ident, pose="frontal"):
    """Render one face patch. Returns (rgb float32 [s,s,3] in 0..255, alpha float32 [s,s])."""
    p = identity_params(ident) if isinstance(ident, (int, np.integer)) else ident
    s = int(size)
    g = (np.arange(s, dtype=np.float64) + 0.5) / s
    u, v = np.meshgrid(g, g)
    shift = 0.0
    if pose == "roll_left" or pose == "roll_right":
        a = np.deg2rad(14.0) * (1 if pose == "roll_left" else -1)
        du, dv = u - 0.5, v - 0.5
        u = 0.5 + np.cos(a) * du - np.sin(a) * dv
        v = 0.5 + np.sin(a) * du + np.cos(a) * dv
    elif pose == "yaw_left":
        shift = -0.07
    elif pose == "yaw_right":
        shift = 0.07

    def ell(cx, cy, rx, ry):
        return _cover(((u - cx) / rx) ** 2 + ((v - cy) / ry) ** 2, min(rx, ry) * s)

    img = np.empty((s, s, 3), np.float64)
    img[:] = p["skin"]
    alpha = ell(0.5, 0.5, 0.42, 0.49)
    # cheeks + identity mark
    for cx in (0.30 + shift, 0.70 + shift):
        m = ell(cx, 0.60, 0.09, 0.07)[..., None] * 0.6
        img = img * (1 - m) + p["cheek"] * m
    m = ell(p["mark_pos"][0] + shift, p["mark_pos"][1], p["mark_r"], p["mark_r"])[..., None]
    img = img * (1 - m) + p["mark_col"] * m
    # hair: top band of the head ellipse
    hair = np.clip((p["hair_frac"] + 0.02 - v) * s * 0.5 + 0.5, 0, 1)[..., None]
    img = img * (1 - hair) + p["hair"] * hair
    dark = np.array([25.0, 20.0, 20.0])
    # brows
    for cx in (0.5 - p["eye_dx"] + shift, 0.5 + p["eye_dx"] + shift):
        m = ell(cx, 0.31, 0.11, 0.018)[..., None]
        img = img * (1 - m) + dark * m
    # eyes: white + dark iris
    for cx in (0.5 - p["eye_dx"] + shift, 0.5 + p["eye_dx"] + shift):
        m = ell(cx, 0.40, p["eye_r"][0], p["eye_r"][1])[..., None]
        img = img * (1 - m) + np.array([235.0, 235.0, 235.0]) * m
        m = ell(cx + shift * 0.5, 0.40, p["eye_r"][1] * 0.95, p["eye_r"][1] * 0.95)[..., None]
        img = img * (1 - m) + dark * m
    # nose
    m = ell(0.5 + shift * 1.3, 0.52, 0.028, 0.10)[..., None] * 0.55
    img = img * (1 - m) + (p["skin"] * 0.55) * m
    m = ell(0.5 + shift * 1.3, 0.62, 0.07, 0.022)[..., None] * 0.7
    img = img * (1 - m) + dark * m
    # mouth
    m = ell(0.5 + shift, 0.74, p["mouth_r"][0], p["mouth_r"][1])[..., None]
    img = img * (1 - m) + p["mouth_col"] * m
    m = ell(0.5 + shift, 0.74, p["mouth_r"][0] * 0.8, p["mouth_r"][1] * 0.25)[..., None]
    img = img * (1 - m) + dark * m
    return img.astype(np.float32), alpha.astype(np.float32)


def lowpass_noise(rng, h, w, cell=32, sigma=12.0):
    """smooth background: coarse gaussian noise, bilinearly upsampled"""
    gh, gw = h // cell + 2, w // cell + 2
    coarse = rng.normal(0, sigma, (gh, gw, 3))
    ys = (np.arange(h) + 0.5) / cell
    xs = (np.arange(w) + 0.5) / cell
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    a = coarse[y0][:, x0]; b = coarse[y0][:, x0 + 1]
    c = coarse[y0 + 1][:, x0]; d = coarse[y0 + 1][:, x0 + 1]
    return (1 - fy) * ((1 - fx) * a + fx * b) + fy * ((1 - fx) * c + fx * d)


class SyntheticVideo(object):
    """Synthetic clip with the `Video` iteration contract of the reference (video.py:411-464)."""

    def __init__(self, width=1920, height=1080, n_frames=1000, frame_rate=25.0, n_shots=4, faces=8,
                 identities=12, min_face=80, max_face=240, seed=20260925, noise=3.0):
        self.frame_rate = float(frame_rate)
        self._size = (int(width), int(height))
        self._frame_size = self._size
        self.n_frames = int(n_frames)
        self.duration = self.n_frames / self.frame_rate
        self.step, self.start, self.end = 1.0 / self.frame_rate, 0.0, self.duration      # what structure.Shot reads (video.py:186-190)
        self.n_shots = int(n_shots)
        self.faces = int(faces)
        self.seed = seed
        rng = np.random.default_rng(seed)
        w, h = self._size
        # shot boundaries (frame indices), equal length
        self.shot_bounds = [int(round(k * self.n_frames / self.n_shots)) for k in range(self.n_shots + 1)]
        # noise tile
        self._noise = np.clip(np.rint(rng.normal(0, noise, (h + 64, w + 64, 1))), -127, 127).astype(np.int16)
        self._noise_off = rng.integers(0, 64, (self.n_frames, 2))
        # region grid for faces
        cols = int(np.ceil(np.sqrt(self.faces * w / float(h))))
        rows = int(np.ceil(self.faces / float(cols)))
        self._bg = []
        self.tracks = []   # per shot: list of dict(ident, pose, cx[], cy[], size[])
        max_face = int(min(max_face, 0.9 * min(w / cols, h / rows)))
        min_face = int(min(min_face, max_face))
        for sidx in range(self.n_shots):
            mean = rng.uniform(60, 180)
            bg = lowpass_noise(rng, h, w) + mean
            self._bg.append(np.clip(np.rint(bg), 0, 255).astype(np.int16))
            n = self.shot_bounds[sidx + 1] - self.shot_bounds[sidx]
            idents = rng.choice(identities, size=self.faces, replace=self.faces > identities)
            tr = []
            for f in range(self.faces):
                ry, rx = divmod(f, cols)
                x0, x1 = rx * w / cols, (rx + 1) * w / cols
                y0, y1 = ry * h / rows, (ry + 1) * h / rows
                s0 = rng.uniform(min_face, max_face)
                amp = rng.uniform(0, min(10.0, 0.08 * s0))
                period = rng.uniform(80, 200)
                tt = np.arange(n)
                size = np.clip(np.rint(s0 + amp * np.sin(2 * np.pi * tt / period)), min_face, max_face).astype(int)
                # smooth random walk: low-pass filtered steps, <= ~3 px/frame
                step = rng.normal(0, 1.0, (n, 2))
                kl = max(1, min(15, n))
                k = np.ones(kl) / float(kl)
                step = np.stack([np.convolve(step[:, 0], k, "same"), np.convolve(step[:, 1], k, "same")], 1) * 4.0
                pos = np.cumsum(np.clip(step, -3, 3), 0)
                half = size / 2.0 + 2
                cx = (x0 + x1) / 2 + pos[:, 0]
                cy = (y0 + y1) / 2 + pos[:, 1]
                cx = np.clip(cx, x0 + half, x1 - half)
                cy = np.clip(cy, y0 + half, y1 - half)
                rng.integers(0, len(POSES))   # (keeps the random stream of earlier revisions)
                # the pose is part of the identity: the synthetic landmark model is box-relative, so a different pose of the
                # same person would yield a differently aligned chip -- something real landmarks would undo
                tr.append({"ident": int(idents[f]), "pose": POSES[int(idents[f]) % len(POSES)],
                           "cx": cx, "cy": cy, "size": size})
            self.tracks.append(tr)
        self._patch_cache = {}

    # ---- reference Video surface -----------------------------------------------------------
    @property
    def size(self):
        return self._size

    @property
    def frame_size(self):
        return self._frame_size

    @frame_size.setter
    def frame_size(self, value):
        # the reference's Video resizes every frame it hands out to this size (video.py:180-187,402-403); this source keeps rendering
        # at native size and leaves the resize to the consumer (TrackingByDetection does it on the device)
        self._frame_size = tuple(int(v) for v in value)

    def __len__(self):
        return self.n_frames

    def timestamp(self, i):
        return i / self.frame_rate

    def __iter__(self):
        for i in range(self.n_frames):
            yield self.timestamp(i), self.frame(i)

    def shots(self):
        """list of (start, end) seconds -- what `shots.json` holds (reference pyannote-face.py:253-257)"""
        out = []
        for k in range(self.n_shots):
            out.append((self.shot_bounds[k] / self.frame_rate, self.shot_bounds[k + 1] / self.frame_rate))
        return out

    # ---- rendering ---------------------------------------------------------------------------
    def _patch(self, ident, pose, size):
        key = (ident, pose, int(size))
        if key not in self._patch_cache:
            rgb, a = render_face(size, ident, pose)
            self._patch_cache[key] = (rgb, a[..., None])
        return self._patch_cache[key]

    def shot_of(self, i):
        for k in range(self.n_shots):
            if i < self.shot_bounds[k + 1]:
                return k
        return self.n_shots - 1

    def face_boxes(self, i):
        """ground truth [(ident, l, t, r, b)] of frame i (box = rendered patch)"""
        k = self.shot_of(i)
        j = i - self.shot_bounds[k]
        out = []
        for tr in self.tracks[k]:
            s = int(tr["size"][j])
            l = int(np.floor(tr["cx"][j] - s / 2.0)); t = int(np.floor(tr["cy"][j] - s / 2.0))
            out.append((tr["ident"], l, t, l + s - 1, t + s - 1))
        return out

    def paste_list(self, i):
        """[(l, t, rgb float32 [s,s,3], alpha float32 [s,s,1])] for frame i"""
        k = self.shot_of(i)
        j = i - self.shot_bounds[k]
        out = []
        for tr in self.tracks[k]:
            s = int(tr["size"][j])
            l = int(np.floor(tr["cx"][j] - s / 2.0)); t = int(np.floor(tr["cy"][j] - s / 2.0))
            rgb, a = self._patch(tr["ident"], tr["pose"], s)
            out.append((l, t, rgb, a))
        return out

    def frame(self, i):
        w, h = self._size
        k = self.shot_of(i)
        img = self._bg[k].astype(np.float32)
        for l, t, rgb, a in self.paste_list(i):
            s = rgb.shape[0]
            sub = img[t:t + s, l:l + s]
            sub[...] = np.rint(rgb * a + sub * (1 - a))
        oy, ox = self._noise_off[i]
        out = img.astype(np.int16) + self._noise[oy:oy + h, ox:ox + w]
        return np.ascontiguousarray(np.clip(out, 0, 255).astype(np.uint8))

    # ---- HBM-resident generation (bench): identical bytes to frame(i), integer ops only on the device ------------------
    def frames_torch(self, device, indices=None):
        """uint8 tensor [n, H, W, 3] on `device`, bit-identical to np.stack([self.frame(i)]).  The per-face alpha blend
        is done on the host on the small patch (float32, same expression as frame()); the full-frame work (background copy,
        paste, noise add, clamp) is int16 arithmetic on the device."""
        import torch
        w, h = self._size
        idx = list(range(self.n_frames)) if indices is None else list(indices)
        out = torch.empty((len(idx), h, w, 3), dtype=torch.uint8, device=device)
        bg_dev = [torch.from_numpy(b).to(device) for b in self._bg]
        noise_dev = torch.from_numpy(self._noise).to(device)
        for n, i in enumerate(idx):
            k = self.shot_of(i)
            img = bg_dev[k].clone()
            bg = self._bg[k]
            for l, t, rgb, a in self.paste_list(i):
                s = rgb.shape[0]
                sub = bg[t:t + s, l:l + s].astype(np.float32)
                patch = np.rint(rgb * a + sub * (1 - a)).astype(np.int16)
                img[t:t + s, l:l + s] = torch.from_numpy(patch).to(device)
            oy, ox = self._noise_off[i]
            img = img + noise_dev[oy:oy + h, ox:ox + w]
            out[n] = img.clamp_(0, 255).to(torch.uint8)
        return out
