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


# ---------------------------------------------------------------------------------------------
# dlib `.dat` files (dlib::serialize streams).  The reference hands dlib model files to the library by path
# (README.md:29-30, face.py:58,62, scripts/pyannote-face.py:37,451-452):
#     shape_predictor_68_face_landmarks.dat          -> dlib.shape_predictor
#     dlib_face_recognition_resnet_model_v1.dat      -> dlib.face_recognition_model_v1
# [EXT] The stream layout below restates dlib's serialize.h / shape_predictor.h / dnn/{core,layers,tensor}.h from their
# published form; dlib's source and the real files are not available in this environment, so the reader is exercised by
# round-tripping through the writer below (tests/test_dlib_dat.py) and against the C++ reader of libpvface (csrc/dlibdat.hip).
#
#   integer      1 control byte (low nibble = n payload bytes, 0x80 = negative) + n little-endian magnitude bytes
#   float/double two integers: mantissa (int64) and exponent (int16), value = mantissa * 2**exponent
#                (exponent 32000 / 32001 / 32002 = +inf / -inf / nan)
#   matrix<T>    integers -nr, -nc, then nr*nc elements row-major
#   std::vector  integer size, then the items;   std::string  integer size, then the bytes
#   tensor       int version(2), 4 integers (n, k, nr, nc), then n*k*nr*nc raw little-endian IEEE floats
class DlibWriter(object):
    def __init__(self):
        self.parts = []

    def int(self, v):
        v = int(v)
        neg = 0x80 if v < 0 else 0
        v = abs(v)
        payload = bytearray()
        while True:
            payload.append(v & 0xFF)
            v >>= 8
            if v == 0:
                break
        self.parts.append(bytes([len(payload) | neg]) + bytes(payload))

    def float(self, v, digits=24):
        v = float(np.float32(v)) if digits == 24 else float(v)      # a C++ float has 24 significant bits: the mantissa is exact
        if np.isinf(v):
            self.int(0); self.int(32000 if v > 0 else 32001); return
        if np.isnan(v):
            self.int(0); self.int(32002); return
        m, e = np.frexp(v)
        mant = int(m * float(1 << digits))
        exp = int(e) - digits
        for _ in range(8):
            if mant & 0xFF or mant == 0:
                break
            mant >>= 8
            exp += 8
        self.int(mant); self.int(exp)

    def double(self, v):
        self.float(v, digits=53)

    def string(self, s):
        b = s.encode() if isinstance(s, str) else bytes(s)
        self.int(len(b)); self.parts.append(b)

    def matrix_f32(self, a):
        a = np.asarray(a, np.float32)
        a = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(-1, 1)
        self.int(-a.shape[0]); self.int(-a.shape[1])
        for v in a.reshape(-1):
            self.float(v)

    def tensor(self, a):
        a = np.ascontiguousarray(a, np.float32)
        dims = list(a.shape) + [1] * (4 - a.ndim) if a.size else [0, 0, 0, 0]
        self.int(2)
        for d in dims:
            self.int(d)
        self.parts.append(a.astype("<f4").tobytes())

    def bytes(self):
        return b"".join(self.parts)


class DlibReader(object):
    def __init__(self, data):
        self.b = memoryview(data)
        self.o = 0

    def int(self):
        c = self.b[self.o]
        n, neg = c & 0x0F, c & 0x80
        if n == 0 or n > 8 or (c & 0x70):
            raise IOError("dlib stream: bad integer control byte 0x%02x at offset %d" % (c, self.o))
        v = int.from_bytes(self.b[self.o + 1:self.o + 1 + n], "little")
        self.o += 1 + n
        return -v if neg else v

    def float(self):
        m, e = self.int(), self.int()
        if e == 32000:
            return float("inf")
        if e == 32001:
            return float("-inf")
        if e == 32002:
            return float("nan")
        return float(np.ldexp(float(m), e))

    def string(self):
        n = self.int()
        s = bytes(self.b[self.o:self.o + n])
        self.o += n
        return s

    def matrix_f32(self):
        nr, nc = -self.int(), -self.int()
        if nr < 0 or nc < 0:
            raise IOError("dlib stream: matrix header expected at offset %d" % self.o)
        out = np.empty(nr * nc, np.float32)
        for i in range(nr * nc):
            out[i] = self.float()
        return out.reshape(nr, nc)

    def tensor(self):
        ver = self.int()
        if ver != 2:
            raise IOError("dlib stream: tensor version %d" % ver)
        dims = [self.int() for _ in range(4)]
        n = int(np.prod(dims))
        a = np.frombuffer(self.b[self.o:self.o + 4 * n], "<f4").astype(np.float32).reshape(dims)
        self.o += 4 * n
        return a


def _unpack_many(rd, count):
    """`count` (mantissa, exponent) floats -> float32 array; the bulk of a shape predictor file (65 M leaf values)"""
    out = np.empty(count, np.float32)
    b, o = rd.b, rd.o
    ldexp = np.ldexp
    for i in range(count):
        c = b[o]; n = c & 0x0F
        m = int.from_bytes(b[o + 1:o + 1 + n], "little")
        if c & 0x80:
            m = -m
        o += 1 + n
        c = b[o]; n = c & 0x0F
        e = int.from_bytes(b[o + 1:o + 1 + n], "little")
        if c & 0x80:
            e = -e
        o += 1 + n
        out[i] = ldexp(float(m), e) if e < 32000 else (np.inf if e == 32000 else (-np.inf if e == 32001 else np.nan))
    rd.o = o
    return out


def write_dlib_shape_predictor(path, model):
    """our tensors -> dlib::shape_predictor stream: int version(1); matrix<float,0,1> initial_shape;
    vector<vector<regression_tree>> forests (tree = vector<split_feature{idx1, idx2, thresh}>, vector<matrix<float,0,1>> leaves);
    vector<vector<unsigned long>> anchor_idx; vector<vector<vector<float,2>>> deltas"""
    meta = model["sp.meta"]
    nc, nt = int(meta[0]), int(meta[1])
    w = DlibWriter()
    w.int(1)
    w.matrix_f32(model["sp.initial_shape"].reshape(-1, 1))
    i1, i2, th, lv = model["sp.split_idx1"], model["sp.split_idx2"], model["sp.split_thresh"], model["sp.leaves"]
    w.int(nc)
    for c in range(nc):
        w.int(nt)
        for t in range(nt):
            w.int(i1.shape[2])
            for s in range(i1.shape[2]):
                w.int(i1[c, t, s]); w.int(i2[c, t, s]); w.float(th[c, t, s])
            w.int(lv.shape[2])
            for l in range(lv.shape[2]):
                w.matrix_f32(lv[c, t, l].reshape(-1, 1))
    w.int(nc)
    for c in range(nc):
        a = model["sp.anchor_idx"][c]
        w.int(len(a))
        for v in a:
            w.int(v)
    w.int(nc)
    for c in range(nc):
        d = model["sp.deltas"][c]
        w.int(len(d))
        for x, y in d:
            w.float(x); w.float(y)
    with open(path, "wb") as f:
        f.write(w.bytes())


def read_dlib_shape_predictor(path):
    """dlib::shape_predictor stream -> the tensors of our container (see write_dlib_shape_predictor for the layout)"""
    with open(path, "rb") as f:
        rd = DlibReader(f.read())
    ver = rd.int()
    if ver != 1:
        raise IOError("%s: shape_predictor version %d (1 expected)" % (path, ver))
    initial = rd.matrix_f32().reshape(-1)
    nc = rd.int()
    forests = []
    for _ in range(nc):
        nt = rd.int()
        trees = []
        for _ in range(nt):
            ns = rd.int()
            sp = np.empty((ns, 3), np.float64)
            for s in range(ns):
                sp[s, 0] = rd.int(); sp[s, 1] = rd.int(); sp[s, 2] = rd.float()
            nl = rd.int()
            leaves = []
            for _ in range(nl):
                nr, ncol = -rd.int(), -rd.int()
                leaves.append(_unpack_many(rd, nr * ncol))
            trees.append((sp, np.stack(leaves)))
        forests.append(trees)
    anchors = []
    for _ in range(rd.int()):
        anchors.append(np.array([rd.int() for _ in range(rd.int())], np.int32))
    deltas = []
    for _ in range(rd.int()):
        n = rd.int()
        deltas.append(_unpack_many(rd, 2 * n).reshape(n, 2))
    nt, ns, nl = len(forests[0]), forests[0][0][0].shape[0], forests[0][0][1].shape[0]
    depth = int(round(np.log2(nl)))
    if (1 << depth) != nl or ns != nl - 1:
        raise IOError("%s: regression trees are not complete binary trees" % path)
    return {
        "sp.meta": np.array([nc, nt, len(initial) // 2, len(anchors[0]), depth], np.int32),
        "sp.initial_shape": initial.astype(np.float32),
        "sp.anchor_idx": np.stack(anchors).astype(np.int32),
        "sp.deltas": np.stack(deltas).astype(np.float32),
        "sp.split_idx1": np.array([[t[0][:, 0] for t in trees] for trees in forests]).astype(np.int32),
        "sp.split_idx2": np.array([[t[0][:, 1] for t in trees] for trees in forests]).astype(np.int32),
        "sp.split_thresh": np.array([[t[0][:, 2] for t in trees] for trees in forests]).astype(np.float32),
        "sp.leaves": np.array([[t[1] for t in trees] for trees in forests]).astype(np.float32),
    }


# [EXT] dlib's mean_face_shape_x / _y (image_transforms/interpolation.h, get_face_chip_details): compiled into dlib, not stored
# in the model file, so a `.dat` embedder uses these recalled constants (51 points = landmarks 17..67).
DLIB_MEAN_FACE_X = [
    0.000213256, 0.0752622, 0.18113, 0.29077, 0.393397, 0.586856, 0.689483, 0.799124, 0.904991, 0.98004, 0.490127, 0.490127,
    0.490127, 0.490127, 0.36688, 0.426036, 0.490127, 0.554217, 0.613373, 0.121737, 0.187122, 0.265825, 0.334606, 0.260918,
    0.182743, 0.645647, 0.714428, 0.793132, 0.858516, 0.79751, 0.719335, 0.254149, 0.340985, 0.428858, 0.490127, 0.551395,
    0.639268, 0.726104, 0.642159, 0.556721, 0.490127, 0.423532, 0.338094, 0.290379, 0.428096, 0.490127, 0.552157, 0.689874,
    0.553364, 0.490127, 0.42689]
DLIB_MEAN_FACE_Y = [
    0.106454, 0.038915, 0.0187482, 0.0344891, 0.0773906, 0.0773906, 0.0344891, 0.0187482, 0.038915, 0.106454, 0.203352,
    0.307009, 0.409805, 0.515625, 0.587326, 0.609345, 0.628106, 0.609345, 0.587326, 0.216423, 0.178758, 0.179852, 0.231733,
    0.245099, 0.244077, 0.231733, 0.179852, 0.178758, 0.216423, 0.244077, 0.245099, 0.780233, 0.745405, 0.727388, 0.742578,
    0.727388, 0.745405, 0.780233, 0.864805, 0.902192, 0.909281, 0.902192, 0.864805, 0.784792, 0.778746, 0.785343, 0.778746,
    0.784792, 0.824182, 0.831803, 0.824182]

# Layer walk of dlib's anet_type (face_recognition_model_v1), input side first -- the order in which a network stream holds the
# layer `details` (add_layer serialises its sub-network before its own details):
#   con(7x7/2) affine relu max_pool, then per residual unit: con affine relu con affine [avg_pool on down units] add_prev relu ...,
#   avg_pool_everything, fc_no_bias(128), loss_metric
def write_dlib_embedder(path, model):
    """our blob -> a dlib network stream with anet_type's layer order.  Layer details are written exactly as dlib's layers
    write them (tag string, params tensor, hyper-parameters); the add_layer bookkeeping around them (version, setup flags,
    empty gradient / output tensors) follows dnn/core.h [EXT]."""
    p = split_resnet_blob(model["emb.blob"])
    empty = np.zeros((0,), np.float32)

    def layer_tail(w):
        w.int(1); w.int(1); w.int(0)          # this_layer_setup_called, gradient_input_is_stale, get_output_and_gradient_input_disabled
        w.tensor(empty); w.tensor(empty)      # x_grad, cached_output
        w.tensor(empty)                       # params_grad (version 2)

    details = []    # innermost first

    def affine(g, b):
        def emit(w, g=g, b=b):
            w.string("affine_")
            w.tensor(np.concatenate([g.reshape(-1), b.reshape(-1)]))
            w.int(1)                                                              # mode = CONV_MODE
        details.append(emit)

    def simple(tag, ints=()):
        def emit(w, tag=tag, ints=ints):
            w.string(tag)
            for v in ints:
                w.int(v)
        details.append(emit)

    def con_named(wname, bname, k, stride, pad):
        wgt, bias = p[wname], p[bname]
        def emit(w, wgt=wgt, bias=bias):
            w.string("con_4")
            w.tensor(np.concatenate([wgt.reshape(-1), bias.reshape(-1)]))
            w.int(wgt.shape[0]); w.int(k); w.int(k); w.int(stride); w.int(stride); w.int(pad); w.int(pad)
            w.double(1); w.double(1); w.double(1); w.double(0)
        details.append(emit)

    details.clear()
    con_named("conv1.w", "conv1.b", 7, 2, 0)
    affine(p["aff1.g"], p["aff1.b"])
    simple("relu_")
    simple("max_pool_2", (3, 3, 2, 2, 0, 0))
    for u, (cin, n, down) in enumerate(RESNET_UNITS):
        con_named("u%d.a.w" % u, "u%d.a.b" % u, 3, 2 if down else 1, 0 if down else 1)
        affine(p["u%d.a.g" % u], p["u%d.a.beta" % u])
        simple("relu_")
        con_named("u%d.b.w" % u, "u%d.b.b" % u, 3, 1, 1)
        affine(p["u%d.b.g" % u], p["u%d.b.beta" % u])
        if down:
            simple("avg_pool_2", (2, 2, 2, 2, 0, 0))
        simple("add_prev_")
        simple("relu_")
    simple("avg_pool_2", (0, 0, 1, 1, 0, 0))
    def fc(w):
        w.string("fc_2")
        w.int(128); w.int(0)                  # num_outputs, bias_mode = FC_NO_BIAS
        w.tensor(p["fc.w"])
        w.double(1); w.double(1); w.double(1); w.double(0)
    details.append(fc)

    w = DlibWriter()
    w.int(1)                                  # add_loss_layer version
    w.string("loss_metric_2"); w.float(0.04); w.float(0.6)
    # nested add_layer records: one version integer per layer on the way in, details + bookkeeping on the way out
    for _ in details:
        w.int(2)
    w.string("input_rgb_image_sized"); w.float(122.782); w.float(117.001); w.float(104.298); w.int(150); w.int(150)
    for emit in details:
        emit(w)
        layer_tail(w)
    with open(path, "wb") as f:
        f.write(w.bytes())


def _find_all(buf, pat):
    out, o = [], buf.find(pat)
    while o >= 0:
        out.append(o)
        o = buf.find(pat, o + 1)
    return out


def read_dlib_embedder(path):
    """dlib network stream of anet_type -> our container tensors.

    The nesting records around the layers (add_layer / add_tag_layer / add_skip_layer versions and flags) differ between
    dlib releases, so the reader does not depend on them: it locates the self-delimiting `details` records of the layers
    that carry parameters by their length-prefixed tag strings ("con_N", "affine_", "fc_N"), which appear input side first,
    and reads the `params` tensor that follows each tag.  29 con + 29 affine + 1 fc records with anet_type's shapes are
    required; anything else is an error."""
    with open(path, "rb") as f:
        data = f.read()

    def tagged(prefix):
        hits = []
        for o in _find_all(data, prefix.encode()):
            # the tag is a serialised std::string: [0x01][len][bytes]; accept "con_", "con_2" ... "con_9"
            if o < 2 or data[o - 2] != 0x01:
                continue
            ln = data[o - 1]
            if ln < len(prefix) or ln > len(prefix) + 2:
                continue
            tag = data[o:o + ln]
            if not all(48 <= c <= 57 for c in tag[len(prefix):]):
                continue
            hits.append((o - 2, o + ln))
        return hits

    recs = sorted([(s, e, "con") for s, e in tagged("con_")] + [(s, e, "affine") for s, e in tagged("affine_")] +
                  [(s, e, "fc") for s, e in tagged("fc_")])
    cons, affs, fcs = [], [], []
    for s, e, kind in recs:
        rd = DlibReader(data)
        rd.o = e
        try:
            if kind == "fc":
                rd.int(); rd.int()            # num_outputs, bias_mode
            t = rd.tensor()
        except (IOError, IndexError, ValueError):
            continue
        if kind == "con":
            nf = rd.int(); nr = rd.int(); ncol = rd.int(); sy = rd.int(); sx = rd.int()
            cons.append((t.reshape(-1), nf, nr, ncol, sy, sx))
        elif kind == "affine":
            affs.append(t.reshape(-1))
        else:
            fcs.append(t.reshape(-1))
    if len(cons) != 29 or len(affs) != 29 or len(fcs) != 1:
        raise IOError("%s: expected 29 con, 29 affine and 1 fc record (anet_type), found %d / %d / %d" % (path, len(cons), len(affs), len(fcs)))
    lay = resnet_param_layout()
    shapes = dict(lay)
    parts = {}
    conv_names = ["conv1"] + [n for u in range(len(RESNET_UNITS)) for n in ("u%d.a" % u, "u%d.b" % u)]
    for name, (flat, nf, nr, ncol, sy, sx), ab in zip(conv_names, cons, affs):
        wshape = shapes[name + ".w"]
        nw = int(np.prod(wshape))
        if nf != wshape[0] or nr != wshape[2] or ncol != wshape[3] or flat.size != nw + nf or ab.size != 2 * nf:
            raise IOError("%s: layer %s does not have anet_type's shape" % (path, name))
        parts[name + ".w"] = flat[:nw].reshape(wshape)
        parts[name + ".b"] = flat[nw:]
        gname, bname = ("aff1.g", "aff1.b") if name == "conv1" else (name + ".g", name + ".beta")
        parts[gname], parts[bname] = ab[:nf], ab[nf:]
    if fcs[0].size != 256 * 128:
        raise IOError("%s: fc layer is not 256 x 128" % path)
    parts["fc.w"] = fcs[0].reshape(256, 128)
    blob = np.concatenate([np.asarray(parts[name], np.float32).reshape(-1) for name, _ in lay])
    return {
        "emb.meta": np.array([150], np.int32),
        "emb.padding": np.array([0.25], np.float64),
        "emb.mean_shape": np.stack([np.asarray(DLIB_MEAN_FACE_X), np.asarray(DLIB_MEAN_FACE_Y)], 1).astype(np.float32),
        "emb.blob": blob,
    }


def load_model_file(path, kind):
    """tensors of a model file given by path: our `.pvfm` container or a dlib `.dat` stream (kind: 'shape_predictor' | 'embedder')"""
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic == b"PVFMODEL":
        return load_container(path)
    if kind == "shape_predictor":
        return read_dlib_shape_predictor(path)
    if kind == "embedder":
        return read_dlib_embedder(path)
    raise ValueError("unknown model kind %r" % kind)
