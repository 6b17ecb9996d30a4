"""Model files for the face hot path: container I/O and seeded synthetic weights.

The reference loads three dlib models: the HOG detector compiled into dlib
(`dlib.get_frontal_face_detector()`, reference face.py:54), and two files given by path on the CLI
(`shape_predictor_68_face_landmarks.dat`, `dlib_face_recognition_resnet_model_v1.dat`; reference
README.md:29-30, face.py:58,62).  None of them exists in this environment, so this module writes models
of exactly the same *shapes* into a small tagged-tensor container (`*.pvfm`) that the C ABI
(`pvf_load_detector / pvf_load_shape_predictor / pvf_load_embedder`, include/pvface.h) reads.

Container layout (little endian):  b"PVFMODEL" u32 version u32 n  then per tensor:
  u32 name_len, name, u32 dtype(0 f32,1 i32,2 f64,3 u8), u32 ndim, u64 dims[ndim], u64 nbytes, pad to 8, data
"""
import os
import struct
import numpy as np

_DT = {0: np.float32, 1: np.int32, 2: np.float64, 3: np.uint8}
_DTI = {np.dtype(v): k for k, v in _DT.items()}

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
DEFAULT_DETECTOR = os.path.join(DATA_DIR, "frontal_face_detector.pvfm")


def save_container(path, tensors):
    with open(path, "wb") as f:
        f.write(b"PVFMODEL")
        f.write(struct.pack("<II", 1, len(tensors)))
        for name, arr in tensors.items():
            arr = np.ascontiguousarray(arr)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<II", _DTI[arr.dtype], arr.ndim))
            for d in arr.shape:
                f.write(struct.pack("<Q", d))
            f.write(struct.pack("<Q", arr.nbytes))
            pad = (-f.tell()) % 8
            f.write(b"\0" * pad)
            f.write(arr.tobytes())


def load_container(path):
    out = {}
    with open(path, "rb") as f:
        if f.read(8) != b"PVFMODEL":
            raise IOError("%s: not a PVFMODEL container" % path)
        _, n = struct.unpack("<II", f.read(8))
        for _ in range(n):
            (ln,) = struct.unpack("<I", f.read(4))
            name = f.read(ln).decode()
            dt, nd = struct.unpack("<II", f.read(8))
            dims = [struct.unpack("<Q", f.read(8))[0] for _ in range(nd)]
            (nbytes,) = struct.unpack("<Q", f.read(8))
            f.read((-f.tell()) % 8)
            out[name] = np.frombuffer(f.read(nbytes), dtype=_DT[dt]).reshape(dims).copy()
    return out


# ---------------------------------------------------------------------------------------------
# canonical 68-point layout of the synthetic face, in the normalised face box [0,1]^2
# (same index convention as dlib's 68-point model: 0-16 jaw, 17-26 brows, 27-35 nose, 36-47 eyes, 48-67 mouth)
def canonical_shape68():
    p = np.zeros((68, 2), np.float64)
    th = np.pi - np.arange(17) * np.pi / 16
    p[0:17, 0] = 0.5 + 0.42 * np.cos(th)
    p[0:17, 1] = 0.45 + 0.52 * np.sin(th)
    xs = np.linspace(0.20, 0.42, 5)
    p[17:22, 0] = xs
    p[17:22, 1] = 0.30 - 0.03 * np.sin(np.linspace(0, np.pi, 5))
    p[22:27, 0] = 1.0 - xs[::-1]
    p[22:27, 1] = p[17:22, 1][::-1]
    p[27:31, 0] = 0.5
    p[27:31, 1] = np.linspace(0.38, 0.56, 4)
    p[31:36, 0] = np.linspace(0.42, 0.58, 5)
    p[31:36, 1] = 0.62
    for base, cx in ((36, 0.32), (42, 0.68)):
        a = np.pi - np.arange(6) * 2 * np.pi / 6
        p[base:base + 6, 0] = cx + 0.08 * np.cos(a)
        p[base:base + 6, 1] = 0.40 - 0.035 * np.sin(a)
    a = np.pi - np.arange(12) * 2 * np.pi / 12
    p[48:60, 0] = 0.5 + 0.16 * np.cos(a)
    p[48:60, 1] = 0.74 - 0.06 * np.sin(a)
    a = np.pi - np.arange(8) * 2 * np.pi / 8
    p[60:68, 0] = 0.5 + 0.10 * np.cos(a)
    p[60:68, 1] = 0.74 - 0.03 * np.sin(a)
    return p


def mean_face_shape51():
    """Template playing the role of dlib's mean_face_shape_x/y (51 points, landmarks 17..67), normalised so
    that x spans [0,1] ([EXT]: dlib's own constants are not available here; they are data, stored in the model)."""
    p = canonical_shape68()[17:68]
    x0, x1 = p[:, 0].min(), p[:, 0].max()
    y0 = p[:, 1].min()
    q = np.empty_like(p)
    q[:, 0] = (p[:, 0] - x0) / (x1 - x0)
    q[:, 1] = (p[:, 1] - y0) / (x1 - x0)
    return q.astype(np.float32)


# ---------------------------------------------------------------------------------------------
def make_shape_predictor(seed=20260925, n_cascades=15, n_trees=500, n_pix=500, depth=4, leaf_sigma=2e-4):
    """Synthetic ERT with dlib's tensor shapes (15 cascades x 500 trees of depth 4, 500 pixels, 68 parts)."""
    rng = np.random.default_rng(seed)
    n_split, n_leaf, P = (1 << depth) - 1, 1 << depth, 68
    return {
        "sp.meta": np.array([n_cascades, n_trees, P, n_pix, depth], np.int32),
        # the detector's box is the 80x80 window minus its padding cells, ~0.69 of the rendered face patch (like dlib's
        # boxes, which are tighter than the head), so the mean shape is laid out in box coordinates accordingly
        "sp.initial_shape": ((canonical_shape68() - 0.5) * BOX_TO_FACE + 0.5).astype(np.float32).reshape(-1),
        "sp.anchor_idx": rng.integers(0, P, (n_cascades, n_pix), dtype=np.int32),
        "sp.deltas": rng.normal(0, 0.1, (n_cascades, n_pix, 2)).astype(np.float32),
        "sp.split_idx1": rng.integers(0, n_pix, (n_cascades, n_trees, n_split), dtype=np.int32),
        "sp.split_idx2": rng.integers(0, n_pix, (n_cascades, n_trees, n_split), dtype=np.int32),
        "sp.split_thresh": rng.uniform(-40, 40, (n_cascades, n_trees, n_split)).astype(np.float32),
        "sp.leaves": rng.normal(0, leaf_sigma, (n_cascades, n_trees, n_leaf, 2 * P)).astype(np.float32),
    }


BOX_TO_FACE = 1.45

RESNET_UNITS = [(32, 32, 0)] * 3 + [(32, 64, 1)] + [(64, 64, 0)] * 3 + [(64, 128, 1)] + [(128, 128, 0)] * 2 + \
               [(128, 256, 1)] + [(256, 256, 0)] * 2 + [(256, 256, 1)]

# scale of the final fc layer, calibrated once (tools/calibrate_embedder.py) so that different synthetic
# identities land ~1.0 apart and the same identity well under the reference threshold 0.6 (clustering.py:138)
FC_SCALE = 1.3


def resnet_param_layout():
    """[(name, shape)] in blob walk order (see oracle/pvo_resnet.c and csrc/resnet.hip)."""
    lay = [("conv1.w", (32, 3, 7, 7)), ("conv1.b", (32,)), ("aff1.g", (32,)), ("aff1.b", (32,))]
    for u, (cin, n, _) in enumerate(RESNET_UNITS):
        lay += [("u%d.a.w" % u, (n, cin, 3, 3)), ("u%d.a.b" % u, (n,)), ("u%d.a.g" % u, (n,)), ("u%d.a.beta" % u, (n,)),
                ("u%d.b.w" % u, (n, n, 3, 3)), ("u%d.b.b" % u, (n,)), ("u%d.b.g" % u, (n,)), ("u%d.b.beta" % u, (n,))]
    lay += [("fc.w", (256, 128))]
    return lay


def make_embedder(seed=20260926, fc_scale=None):
    """Random-init weights with the architecture of dlib's face_recognition_model_v1 (29 conv layers)."""
    rng = np.random.default_rng(seed)
    parts = []
    for name, shape in resnet_param_layout():
        kind = name.split(".")[-1]
        if kind == "w" and name != "fc.w":
            fan_in = shape[1] * shape[2] * shape[3]
            a = rng.normal(0, np.sqrt(2.0 / fan_in), shape)
        elif name == "fc.w":
            a = rng.normal(0, 1.0 / 16.0, shape) * (FC_SCALE if fc_scale is None else fc_scale)
        elif kind == "b":
            a = rng.normal(0, 0.01, shape)
        elif kind == "g":
            a = (0.5 if ".b.g" in name else 1.0) + rng.normal(0, 0.05, shape)
        else:  # beta
            a = rng.normal(0, 0.05, shape)
        parts.append(a.astype(np.float32).reshape(-1))
    return {
        "emb.meta": np.array([150], np.int32),
        "emb.padding": np.array([0.25], np.float64),
        "emb.mean_shape": mean_face_shape51(),
        "emb.blob": np.concatenate(parts),
    }


def split_resnet_blob(blob):
    out, o = {}, 0
    for name, shape in resnet_param_layout():
        n = int(np.prod(shape))
        out[name] = blob[o:o + n].reshape(shape)
        o += n
    assert o == blob.size
    return out


def dsst_tables():
    """Host-computed constant tables handed to both the HIP tracker and the oracle (no device transcendentals)."""
    r, c = np.mgrid[0:64, 0:64].astype(np.float64)
    dist = np.sqrt((c - 32.0) ** 2 + (r - 32.0) ** 2) / 32.0
    mask64 = np.where(dist < 1, np.cos(dist * np.pi / 2), 0.0)
    ds = np.abs(np.arange(32, dtype=np.float64) - 16.0) / 16.0
    mask_scale = np.where(ds < 1, np.cos(ds * np.pi / 2), 0.0)
    k64 = np.arange(32, dtype=np.float64)
    tw64 = np.stack([np.cos(2 * np.pi * k64 / 64), np.sin(2 * np.pi * k64 / 64)], 1)
    k32 = np.arange(16, dtype=np.float64)
    tw32 = np.stack([np.cos(2 * np.pi * k32 / 32), np.sin(2 * np.pi * k32 / 32)], 1)
    return {
        "mask64": np.ascontiguousarray(mask64), "mask_scale": np.ascontiguousarray(mask_scale),
        "tw64": np.ascontiguousarray(tw64), "tw32": np.ascontiguousarray(tw32),
        "alpha_pow_m16": float(1.02 ** -16.0), "ln_alpha": float(np.log(1.02)),
    }


def ensure_synthetic_models(directory, small=False):
    """Write (once) the synthetic landmark + embedding models; returns (landmark_path, embedding_path)."""
    os.makedirs(directory, exist_ok=True)
    tag = "small" if small else "full"
    lp = os.path.join(directory, "shape_predictor_68_face_landmarks.%s.pvfm" % tag)
    ep = os.path.join(directory, "face_recognition_resnet_model_v1.pvfm")
    if not os.path.exists(lp):
        save_container(lp, make_shape_predictor(n_cascades=3, n_trees=40, n_pix=120) if small else make_shape_predictor())
    if not os.path.exists(ep):
        save_container(ep, make_embedder())
    return lp, ep
