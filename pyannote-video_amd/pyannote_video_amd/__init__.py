"""MI355X-native face hot path of pyannote-video: `Face`, `FaceTracking`, `FaceClustering`, `TrackingByDetection`
with the reference's names and call contracts (pyannote/video/__init__.py:33-44), computing on gfx950 through
libpvface.so (include/pvface.h)."""
from .face import Face
from .face_tracking import FaceTracking
from .tracking_by_detection import TrackingByDetection
from .clustering import FaceClustering
from .synth import SyntheticVideo

__all__ = ['Face', 'FaceTracking', 'TrackingByDetection', 'FaceClustering', 'SyntheticVideo']
