"""On-disk formats that couple the reference's stages (scripts/pyannote-face.py).  Byte-compatible writers and the
parsing rules that decide integer outputs downstream:

  track.txt      "{t:.3f} {identifier:d} {left:.3f} {top:.3f} {right:.3f} {bottom:.3f} {status:s}\\n"  (:116-118,263-267)
  landmarks.txt  "{t:.3f} {identifier:d}" + 68 x " {x/w:.5f} {y/h:.5f}"                                 (:299-305)
  embedding.txt  "{t:.3f} {identifier:d}" + 128 x " {x:.5f}"                                            (:307-311)

`quantise_track_box` is the in-memory equivalent of writing a box to track.txt and reading it back in `extract`
(float32 parse :125-127, multiply by the frame size, int() truncation :142-145)."""
import numpy as np
from . import _lib

FACE_TEMPLATE = ('{t:.3f} {identifier:d} '
                 '{left:.3f} {top:.3f} {right:.3f} {bottom:.3f} '
                 '{status:s}\n')


def track_lines(identifier, track):
    for t, (left, top, right, bottom), status in track:
        yield FACE_TEMPLATE.format(t=t, identifier=identifier, status=status, left=left, right=right, top=top, bottom=bottom)


def write_tracks(path, tracks):
    """tracks: iterable of normalised tracks in yield order; the track id is the enumeration index (:261)"""
    with open(path, 'w') as f:
        for identifier, track in enumerate(tracks):
            for line in track_lines(identifier, track):
                f.write(line)
            f.flush()


def denormalise(box, frame_width, frame_height):
    """float32 normalised box (as parsed from track.txt) -> integer pixel rectangle of `getFaceGenerator` (:142-145).
    The reference iterates `tracking.iterrows()` over a mixed-dtype frame, so pandas hands the float32 columns back as
    Python floats and `left * frame_width` runs in float64 (0.175 * 640 -> 111, not the float32 product 112)."""
    return (int(float(box[0]) * frame_width), int(float(box[1]) * frame_height),
            int(float(box[2]) * frame_width), int(float(box[3]) * frame_height))


def quantise_track_box(box, frame_width, frame_height):
    """(l,t,r,b) normalised -> integer pixel rectangle exactly as `extract` rebuilds it from track.txt"""
    return denormalise([np.float32("%.3f" % v) for v in box], frame_width, frame_height)


def quantise_time(t):
    # == float("%.3f" % t): both are the correctly rounded 3-decimal value (tests/test_host_logic.py) -- for Python floats; round() of a
    # numpy scalar goes through numpy's scale / rint / divide and is not (np.float64(0.1125) -> 0.112), hence the float()
    return round(float(t), 3)


def pandas_sort_order(times):
    """Row order of `tracking.sort_values('t')` (pyannote-face.py:130) for the rows of a track file in file order.

    pandas sorts one float column with numpy's default `argsort(kind='quicksort')`, which is NOT stable: rows that share a
    timestamp (the faces of one frame) come out in an order that depends on the whole column and on numpy's sort kernel for
    this CPU.  That order decides nothing but the order of the faces of one frame in landmarks.txt / embedding.txt; it is
    replayed with the same numpy call so that the files match the reference's line for line on the machine they are made on."""
    return np.argsort(np.asarray(times, np.float64), kind="quicksort")


def read_tracks(path):
    """[(T, identifier, (l,t,r,b) float32 normalised, status)] in the order of getFaceGenerator's sorted table (:121-130)"""
    rows = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            rows.append((float(p[0]), int(p[1]), tuple(np.float32(v) for v in p[2:6]), p[6]))
    return [rows[i] for i in pandas_sort_order([r[0] for r in rows])]


def _row_key(T, identifier):
    """one integer per (3-decimal time, track): a track has at most one row per timestamp"""
    return np.rint(np.asarray(T, np.float64) * 1000.0).astype(np.int64) * (1 << 24) + np.asarray(identifier, np.int64)


def file_order(face_T, face_id, file_T, file_id):
    """permutation that puts extracted faces (any order within a timestamp) into the order the reference's `extract` writes them:
    the order of the (T, track) rows in the pandas-sorted track table.  file_T / file_id: the track file's rows in file order."""
    file_T = np.asarray(file_T, np.float64)
    order = pandas_sort_order(file_T)
    rank_of_row = np.empty(len(order), np.int64)
    rank_of_row[order] = np.arange(len(order))                  # position of every file row in the sorted table
    fkey = _row_key(file_T, file_id)
    sorter = np.argsort(fkey, kind="stable")
    want = _row_key(face_T, face_id)
    pos = np.searchsorted(fkey[sorter], want)
    rows = sorter[np.minimum(pos, len(sorter) - 1)] if len(sorter) else np.zeros(0, np.int64)
    if len(want) and not (len(sorter) and np.array_equal(fkey[rows], want)):
        raise KeyError("a face whose (time, track) is not a row of the track table")
    return np.argsort(rank_of_row[rows], kind="stable")


def landmark_line(T, identifier, pts, frame_width, frame_height):
    s = '{t:.3f} {identifier:d}'.format(t=T, identifier=identifier)
    for x, y in pts:
        s += ' {x:.5f} {y:.5f}'.format(x=int(x) / frame_width, y=int(y) / frame_height)
    return s + '\n'


def embedding_line(T, identifier, embedding):
    s = '{t:.3f} {identifier:d}'.format(t=T, identifier=identifier)
    for x in embedding:
        s += ' {x:.5f}'.format(x=float(x))
    return s + '\n'


def landmark_rows(T, identifier, pts, frame_width, frame_height):
    """bytes of many landmarks.txt lines at once (== landmark_line per face, formatted by the library: pvf_format_rows)"""
    from . import _lib
    pts = np.asarray(pts).reshape(len(T), -1, 2)
    vals = np.stack([pts[:, :, 0].astype(np.float64) / frame_width, pts[:, :, 1].astype(np.float64) / frame_height], 2)
    return _lib.format_rows(T, identifier, vals.reshape(len(T), -1), 5)


def embedding_rows(T, identifier, embeddings):
    """bytes of many embedding.txt lines at once (== embedding_line per face)"""
    from . import _lib
    return _lib.format_rows(T, identifier, np.asarray(embeddings).astype(np.float64).reshape(len(T), 128 if len(T) == 0 else -1), 5)


def quantise_embedding(embedding):
    """float64 values as `preprocess` reads them back from the 5-decimal text (clustering.py:70-74)"""
    return np.array([float('{x:.5f}'.format(x=float(x))) for x in embedding], np.float64)


def read_embeddings(path):
    """-> (time[N], track[N] int, X float64 [N,128]) in file order"""
    with open(path, 'rb') as fp:
        text = fp.read()
    try:
        data = _lib.parse_rows(text)                      # the library's parser: the same float64 values, ~5 x faster than np.loadtxt
    except _lib.PvfError:
        data = np.loadtxt(path, dtype=np.float64, ndmin=2)      # comments, ragged rows: numpy's own error messages
    if data.shape[0] == 0:
        return np.zeros(0), np.zeros(0, np.int64), np.zeros((0, 128))
    return data[:, 0].copy(), data[:, 1].astype(np.int64), np.ascontiguousarray(data[:, 2:])
