/*
 * pvface.h -- C ABI of the MI355X-native face hot path (libpvface.so, built from pyannote-video_amd/csrc).
 *
 * This is the drop-in boundary: every entry point replaces one call the reference makes into dlib / scipy /
 * munkres / pyannote.algorithms.  "ref:" lines cite the reference interface (file:line under /root/reference)
 * the function stands in for.  INTEGRATION.md shows the reference-side ctypes binding.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; pvf_last_error() gives the thread-local message;
 *     nothing throws or aborts across the boundary;
 *   - handles are opaque uint64_t; the caller owns every host buffer, the library owns device memory;
 *   - TWO HIP streams per context, each with the entry points, the scratch buffers and the lock of its side: the DETECTOR side
 *     (pvf_detect, pvf_detect_batch, pvf_detect_many, pvf_frame_resize: pyramid, FHOG and scoring kernels whose grids fill the chip)
 *     and the rest (trackers, chips, landmarks, embedding, clustering, shot detection: latency-bound chains the host waits on,
 *     on a stream of higher priority).  Calls of one side are serialised by the library (they may come from any thread); a call
 *     of the detector side and a call of the other side run side by side, on the host and on the device -- the streaming engine
 *     detects shot k + 1 in one thread while another tracks and extracts shot k.  Frames are read by both sides; a result the
 *     other side consumes (detections -> tracker starts) passes through the host, which is the ordering between the two streams.
 *     Different contexts (= different GPUs / ranks) run side by side in different threads or processes;
 *   - frames are uint8 RGB, HWC, C-contiguous (ref: pyannote/video/video.py:148-149,400-401);
 *   - there is NO CPU fallback: without a gfx950 device pvf_ctx_create fails.
 */
#ifndef PVFACE_H
#define PVFACE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t pvf_handle;
typedef struct { int32_t left, top, right, bottom; } pvf_rect_i32;   /* dlib.rectangle (inclusive ints) */

/* ---- context ------------------------------------------------------------------------------------- */
const char* pvf_last_error(void);
int32_t pvf_version(void);
int32_t pvf_device_count(int32_t* n);
int32_t pvf_ctx_create(int32_t device, pvf_handle* ctx);
/* same, choosing the priority class of the context's main (non-detector) stream: -1 as low as the detector's stream, 0 / +1 above it */
int32_t pvf_ctx_create_prio(int32_t device, int32_t priority_class, pvf_handle* ctx);
int32_t pvf_ctx_destroy(pvf_handle ctx);
int32_t pvf_sync(pvf_handle ctx);

/* ---- models --------------------------------------------------------------------------------------- */
/* ref: face.py:54  dlib.get_frontal_face_detector()  (path == NULL: detector shipped with the package) */
int32_t pvf_load_detector(pvf_handle ctx, const char* path);
/* ref: face.py:58  dlib.shape_predictor(landmarks); README.md:29-30, scripts/pyannote-face.py:37,451-452 pass dlib's own
 * `shape_predictor_68_face_landmarks.dat` by path: accepted as is (dlib::serialize stream), as is the `.pvfm` container */
int32_t pvf_load_shape_predictor(pvf_handle ctx, const char* path);
/* ref: face.py:62  dlib.face_recognition_model_v1(embedding); `dlib_face_recognition_resnet_model_v1.dat` or `.pvfm` */
int32_t pvf_load_embedder(pvf_handle ctx, const char* path);
/* host only (no GPU needed): one named tensor of a model file exactly as the two loaders above parse it
 * (kind 1 = shape predictor, 2 = embedder; names "sp.*" / "emb.*"); out == NULL returns the size in *nbytes */
int32_t pvf_model_tensor(const char* path, int32_t kind, const char* name, void* out, int64_t cap_bytes, int64_t* nbytes);
/* constant tables of dlib.correlation_tracker's default constructor (ref: tracking.py:250), computed once on the host:
 * mask64[64*64], mask_scale[32], tw64[32*2] (cos,sin 2*pi*k/64), tw32[16*2] */
int32_t pvf_set_tracker_tables(pvf_handle ctx, const double* mask64, const double* mask_scale, const double* tw64,
                               const double* tw32, double alpha_pow_m16, double ln_alpha);

/* ---- frames --------------------------------------------------------------------------------------- */
/* stage a host frame into HBM once; detection, trackers, landmarks and embedding all reuse it
 * (the reference re-reads the host array for every dlib call: tracking.py:203,251,426; pyannote-face.py:296-297) */
int32_t pvf_frame_upload(pvf_handle ctx, const uint8_t* rgb, int32_t h, int32_t w, int64_t row_stride_bytes, pvf_handle* frame);
/* `rgb` may also be a device address (a decoder that delivers into HBM): the frame is copied into a buffer of the library's own, so
 * the source may be overwritten as soon as the call returns */
/* wrap a frame that already lives in HBM (no copy; the caller keeps it alive until pvf_frame_release RETURNS: compute calls queue their
 * kernels without waiting for them, so releasing a wrapped frame waits for the compute stream -- only then may the memory be reused) */
int32_t pvf_frame_wrap_device(pvf_handle ctx, const void* dev_rgb, int32_t h, int32_t w, pvf_handle* frame);
/* Releasing never waits for the GPU: the buffer of a frame the library allocated itself (upload, ingest ring, device resize) goes back
 * to a pool together with an event on the compute stream, and whoever takes it next orders its first write behind that event.
 * The frame calls (upload / wrap / release / ingest) may be made from a second thread (a decoder) while another thread runs the
 * compute entry points of the same context; release a frame only after the calls that use it have returned. */
int32_t pvf_frame_release(pvf_handle ctx, pvf_handle frame);
int32_t pvf_frame_release_many(pvf_handle ctx, const pvf_handle* frames, int32_t n);
/* ref: tracking.py:359-362,410-420 -- the reference drops a shot's frame cache when the shot is done; a streaming run here recycles
 * the buffers of released frames, and this call gives pooled buffers beyond `keep_bytes` back to the allocator (*pooled_bytes,
 * optional: what the pool still holds) */
int32_t pvf_frame_pool_trim(pvf_handle ctx, int64_t keep_bytes, int64_t* pooled_bytes);
/* free / total device memory (hipMemGetInfo): the peak-HBM figure of the streaming runs */
int32_t pvf_mem_info(pvf_handle ctx, int64_t* free_bytes, int64_t* total_bytes);
/* device address of a staged frame (to share it with a second context on the same GPU, e.g. a detector stream) */
int32_t pvf_frame_device_ptr(pvf_handle ctx, pvf_handle frame, const void** dev_rgb);

/* ---- frame ingest ------------------------------------------------------------ */
/* ref: video.py:368-406 (one pipe read + numpy array per frame), :402-403 (cv2.resize), pyannote-face.py:261,287 (two decodes).
 * A ring of `depth` pinned host slots of one frame size; the decoder writes a frame into the slot pvf_ingest_acquire hands out
 * (it returns when that slot's previous upload has left the buffer), pvf_ingest_submit queues ONE asynchronous host-to-HBM copy
 * on the ring's own copy stream and returns a frame handle at once: kernels that read the frame wait for its copy on the device,
 * so uploads overlap the detector.  Release the frame with pvf_frame_release (its buffer is recycled). */
int32_t pvf_ingest_create(pvf_handle ctx, int32_t h, int32_t w, int32_t depth, pvf_handle* ring);
int32_t pvf_ingest_destroy(pvf_handle ctx, pvf_handle ring);
int32_t pvf_ingest_acquire(pvf_handle ctx, pvf_handle ring, int32_t* slot, uint8_t** host_rgb);
int32_t pvf_ingest_submit(pvf_handle ctx, pvf_handle ring, int32_t slot, pvf_handle* frame);
int32_t pvf_ingest_wait(pvf_handle ctx, pvf_handle ring);       /* all queued uploads done (measurement, shutdown) */
/* ref: video.py:180-187,402-403 + tracking.py:389-400  cv2.resize(frame, (w, h)) for detection on down-scaled frames (--min-size):
 * OpenCV's 8-bit INTER_LINEAR on the device; the source frame stays resident for `extract` */
int32_t pvf_frame_resize(pvf_handle ctx, pvf_handle frame, int32_t out_w, int32_t out_h, pvf_handle* out);

/* ---- S1 detector ---------------------------------------------------------------------------------- */
/* ref: face.py:64-67  for face in self.face_detector_(rgb, 1)  -> dlib.rectangle list, NMS order.
 * scores (optional) receive dlib's detection_confidence (score - threshold). */
int32_t pvf_detect(pvf_handle ctx, pvf_handle frame, int32_t upsample, double adjust_threshold,
                   pvf_rect_i32* out, float* scores, int32_t cap, int32_t* n);
/* same for many frames of identical size in one launch sequence; counts[i] = detections of frame i,
 * written at out[i*cap_per_frame ...] */
int32_t pvf_detect_batch(pvf_handle ctx, const pvf_handle* frames, int32_t n_frames, int32_t upsample,
                         double adjust_threshold, pvf_rect_i32* out, float* scores, int32_t* counts,
                         int32_t cap_per_frame);
/* many frames of one size, processed `batch` at a time with the host post-processing of a batch hidden behind the kernels of
 * the next one; results identical to pvf_detect_batch on each batch (what the pipeline calls once per shot) */
int32_t pvf_detect_many(pvf_handle ctx, const pvf_handle* frames, int32_t n_frames, int32_t batch, int32_t upsample,
                        double adjust_threshold, pvf_rect_i32* out, float* scores, int32_t* counts, int32_t cap);

/* the detector's screening pass (on by default; csrc/screen.hip).  on = 1: every window is first scored on the f16 matrix cores with
 * a proven error bound, and the exact fp32 chain runs only for the windows whose approximate score comes within the bound of the
 * threshold; on = 0: the exact chain is evaluated for every window (score_roll_k).  The rectangles, their order and their scores are
 * the same bits either way -- a call whose list overflows or whose features exceed the bound's assumption is repeated on the dense
 * kernel.  list_cap > 0 sets the (window, filter) pairs a batch may list (default 1 << 20). */
int32_t pvf_detector_screening(pvf_handle ctx, int32_t on, int32_t list_cap);
/* batches screened, (window, filter) pairs listed for exact scoring, calls repeated on the dense kernel -- since the context was created;
 * bounds[0..n_filters) (may be NULL) = the error bound per filter, in score units; pipe_err (may be NULL) = what the context measured
 * on its device before the first screened batch: worst |matrix pipe - exact| / sum of magnitudes over an accumulation of 3200 terms
 * (-1 before that; the bound allows 3200 x 2^-22 = 7.6e-4).  The environment variable PVF_DETECTOR_SCREENING=0 switches screening off
 * for every context the process creates. */
int32_t pvf_detector_screening_stats(pvf_handle ctx, int64_t* batches, int64_t* listed, int64_t* retries, double* bounds, double* pipe_err);

/* ---- S2 correlation tracker ------------------------------------------------------------------------ */
/* ref: tracking.py:250  dlib.correlation_tracker() */
int32_t pvf_tracker_create(pvf_handle ctx, pvf_handle* trk);
/* n trackers at once / n trackers back to the pool: the batched host path creates and kills a few thousand per shot */
int32_t pvf_tracker_create_many(pvf_handle ctx, int32_t n, pvf_handle* trks);
int32_t pvf_tracker_destroy_many(pvf_handle ctx, const pvf_handle* trks, int32_t n);
/* n new trackers that are exact copies (filters, position) of started trackers without a pending deferred update.  The two
 * passes over a shot (tracking.py:184-259 forward, then backward) start one tracker per detection from the same frame and box:
 * the second pass clones instead of recomputing the same filters.  A clone SHARES its source's filters on the device until either
 * side writes them (start_track, a full update, the commit of a deferred update): only then is the 2.4 MB state copied. */
int32_t pvf_tracker_clone_many(pvf_handle ctx, const pvf_handle* src, int32_t n, pvf_handle* dst);
int32_t pvf_tracker_destroy(pvf_handle ctx, pvf_handle trk);
/* ref: tracking.py:251  tracker.start_track(frame, dlib.drectangle(*detection)) ; box = (l,t,r,b) doubles */
int32_t pvf_tracker_start(pvf_handle ctx, pvf_handle trk, pvf_handle frame, const double box[4]);
/* ref: tracking.py:203  confidence = tracker.update(frame)   (peak-to-sidelobe ratio) */
int32_t pvf_tracker_update(pvf_handle ctx, pvf_handle trk, pvf_handle frame, double* psr);
/* ref: tracking.py:165,231-237  tracker.get_position() -> drectangle */
int32_t pvf_tracker_position(pvf_handle ctx, pvf_handle trk, double box[4]);
/* batched forms (an addition; the per-object calls above keep working): tracker i runs on frames[i] */
int32_t pvf_tracker_start_many(pvf_handle ctx, const pvf_handle* trks, const pvf_handle* frames,
                               const double* boxes /* n*4 */, int32_t n);
int32_t pvf_tracker_update_many(pvf_handle ctx, const pvf_handle* trks, const pvf_handle* frames, int32_t n,
                                double* psr /* n */, double* boxes_out /* n*4, may be NULL */);
/* update() without the model update: same confidence and position as pvf_tracker_update_many, filters untouched.  The batched
 * host path kills most trackers right after their first update (tracking.py:219-224: a tracker matched to a new detection is
 * dropped), so it defers the filter update and only pays for it -- pvf_tracker_commit_many, with the SAME frames -- for the
 * trackers that live on.  After the commit the tracker state is bit-identical to an immediate update.  A tracker with an
 * uncommitted deferred update refuses further updates. */
int32_t pvf_tracker_update_many_deferred(pvf_handle ctx, const pvf_handle* trks, const pvf_handle* frames, int32_t n,
                                         double* psr, double* boxes_out);
int32_t pvf_tracker_commit_many(pvf_handle ctx, const pvf_handle* trks, const pvf_handle* frames, int32_t n);

/* ---- S3 rectangles + association (host; tiny, order-sensitive) --------------------------------------- */
/* ref: tracking.py:129-134 _match on dlib.drectangle (width = r-l; empty -> area 0), :160-168 overlap matrix */
int32_t pvf_overlap_matrix(const double* a, int32_t na, const double* b, int32_t nb, double ratio, double* out);
/* ref: tracking.py:121,172  Munkres().compute(cost) on the square n x n matrix -> column of each row */
/* _associate (tracking.py:136-182) in one call: gated overlaps, square padding, cost = max - overlap, Munkres, and only the pairs
 * with a positive overlap.  trackers / detections: [n][4] doubles (l,t,r,b); det_of_tracker[t] = detection index or -1. */
int32_t pvf_associate(const double* trackers, int32_t n_trackers, const double* detections, int32_t n_detections,
                      double min_overlap_ratio, int32_t* det_of_tracker);
int32_t pvf_munkres(const double* cost, int32_t n, int32_t* row_to_col);

/* ---- S4 landmarks + embedding ---------------------------------------------------------------------- */
/* ref: face.py:69-70  self.shape_predictor_(rgb, face) -> 68 integer points; face i lives on frames[i] */
int32_t pvf_landmarks(pvf_handle ctx, const pvf_handle* frames, const pvf_rect_i32* boxes, int32_t n,
                      int32_t* pts /* n*68*2 */);
/* ref: face.py:73-76  face_recognition_.compute_face_descriptor(rgb, landmarks) -> 128 floats */
int32_t pvf_embed(pvf_handle ctx, const pvf_handle* frames, const int32_t* pts /* n*68*2 */, int32_t n,
                  float* out /* n*128 */);
/* ref: pyannote-face.py:296-297  landmarks = face.get_landmarks(rgb, face); embedding = face.get_embedding(rgb, landmarks) -- both for a
 * batch of faces in one call (same results as pvf_landmarks followed by pvf_embed) */
int32_t pvf_landmarks_embed(pvf_handle ctx, const pvf_handle* frames, const pvf_rect_i32* boxes, int32_t n,
                            int32_t* pts /* n*68*2 */, float* out /* n*128 */);
/* the network alone on ready-made 150x150x3 chips (host buffer), for testing K7 in isolation */
int32_t pvf_embed_chips(pvf_handle ctx, const uint8_t* chips, int32_t n, float* out /* n*128 */);
/* the aligned chips alone (get_face_chip_details + extract_image_chips), for testing K6 in isolation */
int32_t pvf_face_chips(pvf_handle ctx, const pvf_handle* frames, const int32_t* pts, int32_t n, uint8_t* chips /* n*150*150*3 */);

/* ---- S5 clustering --------------------------------------------------------------------------------- */
/* ref: clustering.py:100-112  -squareform(pdist(X,'euclidean')) reduced to the T x T matrix of block means;
 * X float64 [N][dim], rows grouped by track, row_start[T+1]; D float64 [T][T] (positive distances).
 * Like the reference (clustering.py:104-112: itertools.combinations(range(n_clusters), 2), matrix[i, j] = matrix[j, i]) only the
 * track pairs i < j are computed; D[j][i] is a copy of D[i][j], the diagonal is zero. */
int32_t pvf_pair_mean_dist(pvf_handle ctx, const double* X, int32_t N, int32_t dim, const int32_t* row_start,
                           int32_t T, double* D);
/* same with the pair distance chosen: metric 0 = Euclidean (the reference, clustering.py:101), 1 = cosine distance 1 - a.b / (|a| |b|)
 * (BASELINE.json north_star's wording; 128-D rows).  pvf_cluster_dist agglomerates either matrix. */
int32_t pvf_pair_mean_dist_metric(pvf_handle ctx, const double* X, int32_t N, int32_t dim, const int32_t* row_start,
                                  int32_t T, int32_t metric, double* D);
/* ---- the host state machine of one shot, array in / array out (csrc/shotgraph.hip) ----------------------------------------------
 * ref: pyannote/video/tracking.py:184-259 (_track: one pass over a shot), :261-357 (_fix, _fill_gaps, components, (min_t, max_t) sort);
 *      scripts/pyannote-face.py:125-127,142-145,262-266 (the track file's numbers).  No device work: pure host code, callable without a
 *      context.  A pass ("lane") is resumable: the trackers' starts and first updates are issued ahead in bulk (the plan, fed by the
 *      caller), and the lane only comes back for trackers that outlive their first update.
 * direction / node kinds: 1 forward, 2 detection, 3 backward (the reference's status order). */
int32_t pvf_lane_create(int32_t n_frames, const int32_t* det_counts /* per shot frame */, const double* det_boxes /* [sum][4], frame after frame */,
                        int32_t direction, double min_confidence, double min_overlap_ratio, int32_t deferring, pvf_handle* lane);
int32_t pvf_lane_destroy(pvf_handle lane);
/* plan of the PROCESSING frames [p0, p0 + n_p) (backward pass: frame n - 1 - p): has_update[n_p]; for the detections of those frames in
 * processing order the tracker handle and its first update's confidence + position (ignored where has_update is 0) */
int32_t pvf_lane_feed_plan(pvf_handle lane, int32_t p0, int32_t n_p, const uint8_t* has_update, const uint64_t* handles, const double* psr,
                           const double* pos);
/* run until the pass ends or needs the caller: *request 0 done, 1 update (reply_psr / reply_pos [n][4] with the NEXT call), 2 commit,
 * 3 plan wanted from processing frame *plan_from; req_handles / req_frames (shot frame indices) name the trackers of requests 1 and 2 */
int32_t pvf_lane_advance(pvf_handle lane, const double* reply_psr, const double* reply_pos, int32_t n_reply, int32_t* request,
                         uint64_t* req_handles, int32_t* req_frames, int32_t cap, int32_t* n_req, int32_t* plan_from);
int32_t pvf_lane_take_dead(pvf_handle lane, uint64_t* out, int32_t cap, int32_t* n);      /* killed trackers, to be released by the caller */
int32_t pvf_lane_edges(pvf_handle lane, int32_t* n_edges, int32_t* u_frame_kind, double* u_box, int32_t* v_frame_kind, double* v_box,
                       double* conf, int32_t cap);                                         /* add_edge calls of the pass, in order */
/* tracks of the shot from its two finished passes: rows [cap][6] = (frame, l, t, r, b, status code: forwards | detections << 8 |
 * backwards << 16 | error << 24); track k = rows track_start[k] .. track_start[k + 1]; counts beyond the caps: nothing written */
int32_t pvf_shot_tracks(pvf_handle lane_forward, pvf_handle lane_backward, const double* times, int32_t n_frames, double max_gap,
                        int32_t* rows, int32_t cap, int32_t* n_rows, int32_t* track_start, int32_t track_cap, int32_t* n_tracks);
/* file_box = float64(float32(round(box / detection size, 3))), pixel_box = int(file_box * frame size): what `track` writes and `extract` reads */
int32_t pvf_track_rows(const int32_t* boxes, int64_t n, int32_t det_width, int32_t det_height, int32_t width, int32_t height,
                       double* file_box, int32_t* pixel_box);
int32_t pvf_round_decimals(const double* in, int64_t n, int32_t decimals, double* out);    /* Python's round(float, decimals), array form */

/* ref: clustering.py:116-119,138-148  FaceClustering(threshold)(starting_point, features): average-linkage HAC from the
 * track partition, stop when the closest pair's mean distance exceeds `threshold`;
 * labels[t] = smallest track index of t's cluster; merge_log optional [(T-1)*4] = (a, b, dist, new_size) */
/* The two halves of pvf_cluster_tracks for several GPUs sharing one global clustering (dist.py).
 * pvf_pair_mean_dist_rows: the COMPLETE rows [track0, track1) of the T x T matrix of pvf_pair_mean_dist (D: T x T, row-major; other
 *   rows are left untouched) -- entries j > i computed, entries j < i the mirror of D[j][i] (clustering.py:111-112), diagonal 0; what
 *   this entry has returned since round 2 (round 4 silently left the part below the diagonal zero: restored in round 5).  COST: the
 *   mirrored part of row i is column i of the rows above it, so the rows [0, track1) are computed, not only [track0, track1) -- for the
 *   last share of a split that is the whole matrix, O(T^2): a job that splits the work calls pvf_pair_upper_rows (dist.py does).
 * pvf_pair_upper_rows: only the UPPER-TRIANGLE entries D[i][j], i < j, of those rows (entries j <= i written as zeros): the share a
 *   rank contributes when the ranks split the pairs by triangle area -- nothing is computed twice.
 * Then the agglomeration of a complete D (pvf_cluster_dist) or of the assembled upper triangle (pvf_cluster_upper mirrors it first).
 * Every entry is produced by the same chain of additions as in the single call, so row ranges computed by different ranks stitch
 * into the single call's matrix bit for bit. */
int32_t pvf_pair_mean_dist_rows(pvf_handle ctx, const double* X, int32_t N, int32_t dim, const int32_t* row_start, int32_t T,
                                int32_t track0, int32_t track1, double* D);
int32_t pvf_pair_upper_rows(pvf_handle ctx, const double* X, int32_t N, int32_t dim, const int32_t* row_start, int32_t T,
                            int32_t track0, int32_t track1, double* D);
int32_t pvf_cluster_dist(pvf_handle ctx, const double* D, const int32_t* row_start, int32_t T, double threshold,
                         int32_t* labels, double* merge_log, int32_t* n_merges);
/* U: T x T with the entries i < j set (anything below the diagonal is ignored), in host (on_device = 0) or device memory */
int32_t pvf_cluster_upper(pvf_handle ctx, const double* U, int32_t on_device, const int32_t* row_start, int32_t T, double threshold,
                          int32_t* labels, double* merge_log, int32_t* n_merges);
int32_t pvf_cluster_tracks(pvf_handle ctx, const double* X, int32_t N, int32_t dim, const int32_t* row_start,
                           int32_t T, double threshold, int32_t* labels, double* merge_log, int32_t* n_merges);
/* The in-memory path (no embedding.txt in between): the float32 descriptors as pvf_embed returned them -- `emb` is a host or a device
 * address (emb_on_device), n_src rows of 128 floats row_stride_bytes apart.  Row k of the clustering's table is
 * np.round(float64(emb[order[k]]), decimals) -- what the reference reads back from the '%.5f' text (pyannote-face.py:307-311,
 * clustering.py:70-75; decimals < 0: no rounding; order NULL: rows as they are) -- gathered and rounded ON THE DEVICE: one upload of
 * 4 bytes per value (none when the rows are in HBM already, e.g. the gathered rows of a multi-GPU run) instead of host passes over an
 * 8-byte table.  128-D rows.  pvf_pair_upper_rows_f32 writes the (track1 - track0) x T values of its rows compactly into rows_out
 * (host or device memory): the share one rank contributes to the all-gather of D. */
int32_t pvf_cluster_tracks_f32(pvf_handle ctx, const float* emb, int64_t row_stride_bytes, int32_t n_src, int32_t emb_on_device,
                               const int32_t* order, int32_t N, int32_t decimals, const int32_t* row_start, int32_t T, int32_t metric,
                               double threshold, int32_t* labels, double* merge_log, int32_t* n_merges);
int32_t pvf_pair_upper_rows_f32(pvf_handle ctx, const float* emb, int64_t row_stride_bytes, int32_t n_src, int32_t emb_on_device,
                                const int32_t* order, int32_t N, int32_t decimals, const int32_t* row_start, int32_t T,
                                int32_t track0, int32_t track1, double* rows_out, int32_t out_on_device);

/* ---- file formats (host) ------------------------------------------------------------------------------ */
/* ref: scripts/pyannote-face.py:299-311  the lines of landmarks.txt / embedding.txt: "{t:.3f} {identifier:d}" + n_cols x " {v:.<decimals>f}"
 * + newline per row, byte for byte what Python's format writes (both are the correctly rounded decimal).  values [n_rows][n_cols];
 * out needs 64 + 42 * n_cols bytes per row; *written = bytes produced. */
int32_t pvf_format_rows(const double* t, const int64_t* identifier, const double* values, int64_t n_rows, int32_t n_cols,
                        int32_t decimals, char* out, int64_t cap, int64_t* written);

/* ref: scripts/pyannote-face.py:307-311, face/clustering.py:70-75  the float64 values `preprocess` reads back from the 5-decimal text of
 * embedding.txt, computed in memory: out[i] = rint((double)x[i] * 10^decimals) / 10^decimals (== np.round of the float64 value) */
int32_t pvf_round_rows(const float* x, int64_t n, int32_t decimals, double* out);

/* ref: face/clustering.py:70-75 (`read_table` of embedding.txt), scripts/pyannote-face.py:223-233 (track.txt)  whitespace-separated numeric
 * text -> float64 [n_rows][n_cols], row-major, the values strtod returns (one IEEE division for plain decimals of <= 15 digits, strtod for
 * the rest); rows of different lengths or a token that is not a number are errors.  cap = doubles `out` can hold (len / 2 + 1 suffices). */
int32_t pvf_parse_rows(const char* text, int64_t len, double* out, int64_t cap, int64_t* n_rows, int32_t* n_cols);

/* ---- measurement ----------------------------------------------------------------------------------- */
/* HIP-event timing of each kernel family on the context's stream ("pyramid","fhog","score","ert","chip","conv",
 * "dsst","pdist","hac"); off by default */
int32_t pvf_prof_enable(pvf_handle ctx, int32_t on);
int32_t pvf_prof_reset(pvf_handle ctx);
int32_t pvf_prof_get(pvf_handle ctx, const char* family, double* total_ms, int64_t* launches);

/* ---- stage access used by the parity tests ------------------------------------------------------------ */
int32_t pvf_debug_pyramid_level(pvf_handle ctx, pvf_handle frame, int32_t upsample, int32_t level,
                                uint8_t* out, int32_t* h, int32_t* w);               /* out may be NULL: dims only */
/* the image pyramids of a batch of frames and nothing else (the detector's resize chain alone, for measurements); returns when they are built */
int32_t pvf_debug_pyramid_batch(pvf_handle ctx, const pvf_handle* frames, int32_t n, int32_t upsample);
/* features of one pyramid level exactly as the batched detector computes them (all levels per launch); out [fh][fw][32] or NULL */
int32_t pvf_debug_level_features(pvf_handle ctx, pvf_handle frame, int32_t upsample, int32_t level,
                                 float* out, int32_t* fh, int32_t* fw);
int32_t pvf_debug_fhog(pvf_handle ctx, const uint8_t* img, int32_t h, int32_t w, int32_t cell, int32_t pad_r,
                       int32_t pad_c, float* out, int32_t* fh, int32_t* fw);          /* out [fh][fw][32] */
int32_t pvf_debug_detect_raw(pvf_handle ctx, pvf_handle frame, int32_t upsample, double adjust, float* scores,
                             int32_t* meta /* cap*8: filter,level,r,c,l,t,r,b */, int32_t cap, int32_t* n);
/* the same for any number of frames through the BATCHED path the engine runs (pvf_detect_many without its non-maximum suppression,
 * screening pass included): counts[n_frames]; rows [cap][5] = (level, filter, row, column, score bits) frame after frame in the
 * detector's canonical order; *total = all candidates (more than cap: only the first cap rows were written).  What the whole-clip
 * fixtures pin (tests/golden/c2_full.npz: the scanner's candidates before NMS, reference face.py:64-67). */
int32_t pvf_debug_detect_raw_many(pvf_handle ctx, const pvf_handle* frames, int32_t n_frames, int32_t batch, int32_t upsample,
                                  double adjust, int32_t* counts, int32_t* rows, int64_t cap, int64_t* total);
int32_t pvf_debug_extract_chip(pvf_handle ctx, pvf_handle frame, const double rect[4], double cs, double sn,
                               int32_t rows, int32_t cols, uint8_t* out);
int32_t pvf_debug_tracker_state(pvf_handle ctx, pvf_handle trk, double* F, double* A, double* B);

/* ---- f4: shot boundary detection ------------------------------------------------------------------
 * ref: pyannote/video/structure/shot.py:71-73 (_convert: RGB -> gray -> cv2.resize to `width` x `height`), :75-99 (dfd: Farneback flow
 *      with (0.5, 3, 15, 3, 5, 1.1, 0), per-pixel displaced lookup, mean absolute difference).  dfd[i] compares frames i and i + 1
 *      (n - 1 values).  tables = 22 floats (Gaussian g / x g / x^2 g for x = 0..5, then ig11, ig03, ig33, ig55: structure.shot_tables()).
 *      gray_out ([n][height][width] bytes) and flow_out ([n-1][height][width][2] floats) may be NULL.  The reference's default small
 *      image (50 pixels wide) has one pyramid level; a side of 64 pixels or more (Shot(height=...), shot.py:53-60) brings OpenCV's coarser
 *      levels -- up to three: halved while both sides stay >= 32 -- which run in the same launch. */
int32_t pvf_shot_dfd(pvf_handle ctx, const pvf_handle* frames, int32_t n, int32_t width, int32_t height, const float* tables,
                     double* dfd, uint8_t* gray_out, float* flow_out);

#ifdef __cplusplus
}
#endif
#endif
