"""Face tracking -- drop-in for `pyannote.video.face.tracking.FaceTracking` (reference face/tracking.py:36-78)."""
from .face import Face, DLIB_SMALLEST_FACE
from .tracking_by_detection import TrackingByDetection, HipTrackers


def get_face_detect(face):
    """detect_func plug-in: frame -> iterable of (left, top, right, bottom) int tuples (face/tracking.py:36-42)"""
    def face_detect(frame):
        for f in face.iterfaces(frame):
            yield (f.left(), f.top(), f.right(), f.bottom())
    return face_detect


def get_face_detect_batch(face):
    def face_detect_batch(frames):
        return [[(f.left(), f.top(), f.right(), f.bottom()) for f in faces] for faces in face.iterfaces_batch(frames)]
    return face_detect_batch


class FaceTracking(TrackingByDetection):
    """Same parameters and defaults as the reference class (detect_min_size=0, detect_every=0, track_min_confidence=10,
    track_min_overlap_ratio=0.3, track_max_gap=0)."""

    def __init__(self, detect_min_size=0., detect_every=0., track_min_confidence=10., track_min_overlap_ratio=0.3,
                 track_max_gap=0., ctx=None, detect_batch_size=8):
        face = Face(ctx=ctx)
        super(FaceTracking, self).__init__(
            detect_func=get_face_detect(face),
            detect_smallest=DLIB_SMALLEST_FACE,
            detect_min_size=detect_min_size,
            detect_every=detect_every,
            track_min_confidence=track_min_confidence,
            track_min_overlap_ratio=track_min_overlap_ratio,
            track_max_gap=track_max_gap,
            trackers=HipTrackers(face.ctx),
            detect_batch_func=get_face_detect_batch(face),
            detect_batch_size=detect_batch_size)
        self.face = face
