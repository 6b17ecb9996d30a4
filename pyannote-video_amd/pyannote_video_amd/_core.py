"""Minimal stand-ins for the two pyannote.core types the face path touches (pyannote.core is not installed here).

Reference usage: `Segment(start, end)` with truthiness = non-empty (clustering.py:55-57,78), and
`Annotation(modality='face')` filled as `annotation[segment, track] = label` (clustering.py:76-80)."""


class Segment(object):
    __slots__ = ("start", "end")

    def __init__(self, start=0.0, end=0.0):
        self.start, self.end = float(start), float(end)

    def __bool__(self):
        # pyannote.core.Segment: empty when end - start is below its precision (1e-6)
        return (self.end - self.start) > 1e-6

    __nonzero__ = __bool__

    @property
    def duration(self):
        return self.end - self.start if self else 0.0

    def __iter__(self):
        return iter((self.start, self.end))

    def __eq__(self, o):
        return isinstance(o, Segment) and (self.start, self.end) == (o.start, o.end)

    def __hash__(self):
        return hash((self.start, self.end))

    def __lt__(self, o):
        return (self.start, self.end) < (o.start, o.end)

    def __repr__(self):
        return "<Segment(%g, %g)>" % (self.start, self.end)


class Annotation(object):
    """ordered {(segment, track): label}"""

    def __init__(self, uri=None, modality=None):
        self.uri, self.modality = uri, modality
        self._d = {}

    def __setitem__(self, key, label):
        segment, track = key
        self._d[(segment, track)] = label

    def __getitem__(self, key):
        return self._d[tuple(key)]

    def __len__(self):
        return len(self._d)

    def itertracks(self, yield_label=False):
        for (segment, track), label in sorted(self._d.items(), key=lambda kv: (kv[0][0].start, kv[0][0].end, str(kv[0][1]))):
            yield (segment, track, label) if yield_label else (segment, track)

    def labels(self):
        return sorted(set(self._d.values()))

    def get_timeline(self):
        return sorted(set(s for s, _ in self._d))

    def copy(self):
        a = Annotation(self.uri, self.modality)
        a._d = dict(self._d)
        return a

    def rename_labels(self, mapping):
        a = self.copy()
        for k, v in a._d.items():
            a._d[k] = mapping.get(v, v)
        return a

    def __eq__(self, o):
        return isinstance(o, Annotation) and self._d == o._d
