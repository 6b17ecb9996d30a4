"""Face processing -- drop-in for `pyannote.video.face.face.Face` (reference face.py:38-132), backed by the HIP library."""
from . import shim
from . import runtime

DLIB_SMALLEST_FACE = 36   # reference face.py:35


class Face(object):
    """Face detection, 68-point landmarks and 128-D embedding.

    Parameters
    ----------
    landmarks : str, optional     path to the 68-landmark predictor model
    embedding : str, optional     path to the face embedding model
    """

    def __init__(self, landmarks=None, embedding=None, ctx=None):
        super(Face, self).__init__()
        self.ctx = ctx or runtime.default_context()
        self.face_detector_ = shim.get_frontal_face_detector(self.ctx)
        if landmarks is not None:
            self.shape_predictor_ = shim.shape_predictor(landmarks, self.ctx)
        if embedding is not None:
            self.face_recognition_ = shim.face_recognition_model_v1(embedding, self.ctx)

    def iterfaces(self, rgb):
        """Iterate over all detected faces (detector run on the frame upsampled once: face.py:64-67)"""
        for face in self.face_detector_(rgb, 1):
            yield face

    def iterfaces_batch(self, frames):
        """[[rectangle]] for many frames of one size, one launch sequence (an addition to the reference API)"""
        return [[shim.rectangle(*b) for b in boxes] for boxes, _ in self.ctx.detect_batch(frames, 1)]

    def get_landmarks(self, rgb, face):
        return self.shape_predictor_(rgb, face)

    def get_embedding(self, rgb, landmarks):
        return self.face_recognition_.compute_face_descriptor(rgb, landmarks)

    def get_debug(self, image, face, landmarks):
        raise NotImplementedError("debug crops are visualisation (reference face.py:78-87 uses cv2 and an undefined self.size)")

    def __call__(self, rgb, return_landmarks=False, return_embedding=False, return_debug=False):
        """Iterate over all faces; yields face or (face[, landmarks][, embedding]) like the reference (face.py:89-132)."""
        for face in self.iterfaces(rgb):
            if not (return_landmarks or return_embedding or return_debug):
                yield face
                continue
            result = (face, )
            landmarks = self.get_landmarks(rgb, face)
            if return_landmarks:
                result = result + (landmarks, )
            if return_embedding:
                result = result + (self.get_embedding(rgb, landmarks), )
            if return_debug:
                result = result + (self.get_debug(rgb, face, landmarks), )
            yield result
