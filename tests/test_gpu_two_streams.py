"""GPU: the two sides of a context (detector stream / everything else) really are independent.  One thread runs detector calls in a loop,
another runs tracker starts + updates, landmarks + descriptors and a clustering in a loop, on the SAME context; every result must equal
the one the same call returned when it ran alone -- bit for bit.  (Round 4 found the one buffer the two sides shared: the tracker's chip
features went into the detector's feature maps and overwrote their zero border; this is the test that would have caught it as a race
as well.)  The detector runs with a lowered threshold so that the windows reaching into the border are among its candidates."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_detector_beside_tracker_extraction_and_clustering(ctx, small_video, model_paths):
    frames = [ctx.upload(small_video.frame(i)) for i in range(8)]
    adj = -0.80712890625                                   # ~20 000 raw candidates per frame: border windows included

    def detector_work():
        raw = ctx.detect_raw(frames[5], 1, adj)
        boxes = ctx.detect_many(frames, 3, 1, arrays=True)
        return raw, [np.array(b) for b in boxes]

    f0, f1 = frames[0], frames[1]
    dets = ctx.detect(f0, 1)[0]
    dbox = [tuple(float(v) for v in b) for b in dets] * 16      # 48 trackers
    rng = np.random.default_rng(3)
    E = (0.1 * rng.normal(size=(400, 128))).astype(np.float32)
    rs = (np.arange(41) * 10).astype(np.int32)

    def other_work():
        trk = ctx.tracker_create_many(len(dbox))
        ctx.tracker_start_many(trk, [f0] * len(dbox), dbox)
        psr, pos = ctx.tracker_update_many(trk, [f1] * len(dbox))
        ctx.tracker_destroy_many(trk)
        pts, emb = ctx.landmarks_embed([f0] * len(dets), [tuple(int(v) for v in b) for b in dets])
        labels, log = ctx.cluster_tracks_f32(E, None, rs, 0.6)
        return psr.copy(), pos.copy(), pts.copy(), emb.copy(), labels.copy(), log.copy()

    ref_d = detector_work()
    ref_o = other_work()
    assert len(ref_d[0]) > 15000 and min(r[3] for r in ref_d[0]) == 5
    errors = []

    def loop(work, ref, n):
        try:
            for _ in range(n):
                got = work()
                if isinstance(ref[0], list):                                   # detector: raw candidate list + arrays
                    assert got[0] == ref[0]
                    assert all(np.array_equal(a, b) for a, b in zip(got[1], ref[1]))
                else:
                    assert all(np.array_equal(a, b) for a, b in zip(got, ref))
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    ta = threading.Thread(target=loop, args=(detector_work, ref_d, 12))
    tb = threading.Thread(target=loop, args=(other_work, ref_o, 40))
    ta.start(); tb.start(); ta.join(); tb.join()
    if errors:
        raise errors[0]
    for f in frames:
        f.release()
