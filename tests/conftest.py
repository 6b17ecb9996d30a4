import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    from pyannote_video_amd import models
    d = tmp_path_factory.mktemp("models")
    models.ensure_synthetic_models(str(d), small=True)
    return str(d)


@pytest.fixture(scope="session")
def model_paths(model_dir):
    from pyannote_video_amd import models
    return models.ensure_synthetic_models(model_dir, small=True)


@pytest.fixture(scope="session")
def ctx(model_paths):
    from pyannote_video_amd.runtime import Context
    c = Context(device=0, landmarks=model_paths[0], embedding=model_paths[1])
    yield c
    c.close()


@pytest.fixture(scope="session")
def full_model_paths(tmp_path_factory):
    """the FULL landmark model (15 cascades x 500 trees x 500 pixels, what bench.py runs) + the embedder"""
    from pyannote_video_amd import models
    return models.ensure_synthetic_models(str(tmp_path_factory.mktemp("models_full")), small=False)


@pytest.fixture(scope="session")
def ctx_full(full_model_paths):
    from pyannote_video_amd.runtime import Context
    c = Context(device=0, landmarks=full_model_paths[0], embedding=full_model_paths[1])
    yield c
    c.close()


@pytest.fixture(scope="session")
def small_video():
    from pyannote_video_amd import synth
    return synth.SyntheticVideo(width=640, height=360, n_frames=12, n_shots=2, faces=3, min_face=50, max_face=110, seed=7)
