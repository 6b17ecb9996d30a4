"""dlib `.dat` model files (README.md:29-30; face.py:58,62): the Python codec round-trips our synthetic models through dlib's
stream layout, and the C loader of libpvface parses the same bytes into the same tensors (host only: no GPU needed)."""
import numpy as np
import pytest
from pyannote_video_amd import models, _lib


def test_integer_and_float_codec_roundtrip():
    w = models.DlibWriter()
    ints = [0, 1, 255, 256, -1, -300, 2 ** 31 - 1, -(2 ** 40), 2 ** 62]
    flts = [0.0, 1.0, -1.5, 3.1415927410125732, 1e-30, -2.5e20, float(np.float32(0.1)), float("inf"), float("-inf")]
    for v in ints:
        w.int(v)
    for v in flts:
        w.float(v)
    w.float(float("nan"))
    w.string("con_4")
    rd = models.DlibReader(w.bytes())
    assert [rd.int() for _ in ints] == ints
    assert [rd.float() for _ in flts] == [float(np.float32(v)) if np.isfinite(v) else v for v in flts]
    assert np.isnan(rd.float())
    assert rd.string() == b"con_4"
    # documented byte layout: 300 = 0x012C -> control 0x02, 0x2C, 0x01 ; -1 -> 0x81 0x01
    w = models.DlibWriter(); w.int(300); w.int(-1)
    assert w.bytes() == bytes([0x02, 0x2C, 0x01, 0x81, 0x01])
    # 1.0f = mantissa 2^23 shifted down by two zero bytes (0x80, exponent -7) -> 0x01 0x80 | 0x81 0x07
    w = models.DlibWriter(); w.float(1.0)
    assert w.bytes() == bytes([0x01, 0x80, 0x81, 0x07])


@pytest.fixture(scope="module")
def dat_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("dat")
    sp = models.make_shape_predictor(n_cascades=3, n_trees=40, n_pix=120)
    emb = models.make_embedder()
    sp_path, emb_path = str(d / "shape_predictor_68_face_landmarks.dat"), str(d / "dlib_face_recognition_resnet_model_v1.dat")
    models.write_dlib_shape_predictor(sp_path, sp)
    models.write_dlib_embedder(emb_path, emb)
    return sp, emb, sp_path, emb_path


def test_python_reader_roundtrip(dat_files):
    sp, emb, sp_path, emb_path = dat_files
    got = models.load_model_file(sp_path, "shape_predictor")
    for k, v in sp.items():
        assert got[k].dtype == v.dtype and np.array_equal(got[k], v), k
    got = models.load_model_file(emb_path, "embedder")
    assert np.array_equal(got["emb.blob"], emb["emb.blob"])
    assert got["emb.mean_shape"].shape == (51, 2) and int(got["emb.meta"][0]) == 150 and float(got["emb.padding"][0]) == 0.25


def test_c_loader_parses_dat_like_the_container(dat_files, tmp_path):
    sp, emb, sp_path, emb_path = dat_files
    for k, v in sp.items():
        got = _lib.model_tensor(sp_path, 1, k, v.dtype)
        assert np.array_equal(got, v.reshape(-1)), k
    assert np.array_equal(_lib.model_tensor(emb_path, 2, "emb.blob", np.float32), emb["emb.blob"])
    mean = _lib.model_tensor(emb_path, 2, "emb.mean_shape", np.float32).reshape(51, 2)
    assert np.allclose(mean[:, 0], models.DLIB_MEAN_FACE_X) and np.allclose(mean[:, 1], models.DLIB_MEAN_FACE_Y)
    # the container route gives the same tensors through the same entry point
    p = str(tmp_path / "sp.pvfm")
    models.save_container(p, sp)
    assert np.array_equal(_lib.model_tensor(p, 1, "sp.leaves", np.float32), sp["sp.leaves"].reshape(-1))


def test_c_loader_rejects_garbage(tmp_path):
    p = str(tmp_path / "junk.dat")
    open(p, "wb").write(b"\x00" * 64)
    with pytest.raises(_lib.PvfError):
        _lib.model_tensor(p, 1, "sp.meta", np.int32)
    with pytest.raises(_lib.PvfError):
        _lib.model_tensor(p, 2, "emb.blob", np.float32)


def test_c_loader_bounds_what_a_file_may_ask_for(dat_files, tmp_path):
    """a corrupt or crafted model file must be refused while it is parsed: sizes are bounded by the bytes that are there (nothing is
    allocated for a header that promises 2^31 elements), and feature / anchor indices outside the model never reach the device"""
    sp, _, _, _ = dat_files
    # a matrix header of 2^31 x 2^31 elements in a 40-byte file
    w = models.DlibWriter(); w.int(1); w.int(-(2 ** 31)); w.int(-(2 ** 31))
    p = str(tmp_path / "huge.dat")
    open(p, "wb").write(w.bytes() + b"\x01\x01" * 16)
    with pytest.raises(_lib.PvfError, match="matrix header"):
        _lib.model_tensor(p, 1, "sp.meta", np.int32)
    # a split that reads feature pixel n_pix (one past the end), an anchor that names part 68
    for key, bad, what in (("sp.split_idx1", 120, "split feature index"), ("sp.split_idx2", 7000, "split feature index"), ("sp.anchor_idx", 68, "anchor index")):
        broken = {k: v.copy() for k, v in sp.items()}
        broken[key].reshape(-1)[5] = bad
        p = str(tmp_path / ("bad_%s.dat" % key))
        models.write_dlib_shape_predictor(p, broken)
        with pytest.raises(_lib.PvfError, match=what):
            _lib.model_tensor(p, 1, "sp.meta", np.int32)
    # a tensor record whose dimensions multiply past the end of the file
    w = models.DlibWriter(); w.string("con_1"); w.int(2); w.int(2 ** 20); w.int(2 ** 20); w.int(2 ** 20); w.int(1)
    p = str(tmp_path / "tensor.dat")
    open(p, "wb").write(b"\x00\x00" + w.bytes() + b"\x00" * 64)
    with pytest.raises(_lib.PvfError, match="expected 29 con"):          # the record is skipped as a look-alike, the file then lacks its layers
        _lib.model_tensor(p, 2, "emb.blob", np.float32)
