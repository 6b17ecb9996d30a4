/*
 * pvo.h -- CPU ORACLE for the pyannote-video face hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (pyannote-video_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in third-party code that is neither vendored
 * in /root/reference nor installed here (dlib == 19.12, reference setup.py:49; pyannote.algorithms
 * >= 0.8, setup.py:44; munkres >= 1.0.7, setup.py:51).  The reference ships no tests, golden vectors
 * or model files.  Every function below restates the *published* algorithm
 * of the dlib routine named in its comment, anchored on the reference call site (file:line under
 * /root/reference).  Where this container offers an independent implementation (scipy pdist,
 * scipy linear_sum_assignment, numpy FFT, torch CPU conv) tests/ pin the oracle against it.
 *
 * Numerics contract shared with the HIP kernels (so integer outputs can be compared bit-exactly):
 *   - compiled with -ffp-contract=off; fused multiply-adds only where fmaf()/fma() is written;
 *   - no libm transcendentals on the data path (sqrt and division are IEEE-exact on both sides;
 *     exp/cos tables are computed once on the host and handed to both sides);
 *   - every floating-point reduction has ONE stated order, written next to it.
 */
#ifndef PVO_H
#define PVO_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVO_FHOG_PLANES 31
#define PVO_FHOG_STRIDE 32 /* planes padded to 32 floats per cell (plane 31 == 0) */

/* ---------------------------------------------------------------- image ops (pvo_image.c) */
void pvo_resize_bilinear_rgb(const uint8_t* in, int ih, int iw, uint8_t* out, int oh, int ow);
/* cv2.resize(img, (ow, oh)), INTER_LINEAR, uint8 RGB (reference video.py:402-403) */
void pvo_cv_resize_linear_rgb(const uint8_t* in, int ih, int iw, uint8_t* out, int oh, int ow);
void pvo_pyramid_up_dims(int ih, int iw, int* oh, int* ow);
void pvo_pyramid_down6_dims(int ih, int iw, int* oh, int* ow);
void pvo_pyr_down2_dims(int ih, int iw, int* oh, int* ow);
void pvo_pyr_down2_rgb(const uint8_t* in, int ih, int iw, uint8_t* out);

/* chip extraction: dlib chip_details + extract_image_chip(s).
 * rect = (l,t,r,b) doubles (drectangle), rotation given as unit vector (cs,sn) instead of an angle. */
typedef struct {
    double l, t, r, b;
    double cs, sn;
    int rows, cols;
} pvo_chip_details;
void pvo_extract_chip_rgb(const uint8_t* img, int h, int w, const pvo_chip_details* d, uint8_t* chip);
/* plain affine bilinear sampling (dlib transform_image + interpolate_bilinear, black background):
 * out(r,c) = img(m00*c + m01*r + bx , m10*c + m11*r + by) */
void pvo_transform_image_rgb(const uint8_t* img, int h, int w, const double m[4], const double b[2],
                             uint8_t* out, int oh, int ow);

/* ---------------------------------------------------------------- FHOG (pvo_fhog.c) */
void pvo_fhog_dims(int ih, int iw, int cell, int pad_r, int pad_c, int* fh, int* fw);
/* out: [fh][fw][32] floats, zero padded borders, plane 31 = 0 */
void pvo_fhog(const uint8_t* img, int ih, int iw, int cell, int pad_r, int pad_c, float* out);

/* ---------------------------------------------------------------- detector (pvo_detect.c) */
typedef struct {
    int n_filters;      /* 5 for the frontal face detector */
    int frows, fcols;   /* filter size in cells incl. padding (10x10) */
    int cell;           /* 8 */
    int padding;        /* 1 */
    int win_w, win_h;   /* 80x80 */
    int min_layer_w, min_layer_h; /* 64x64 */
    int max_levels;     /* 1000 */
    double nms_iou, nms_covered;
    const float* w;     /* [n_filters][frows][fcols][32] */
    const float* thresh;/* [n_filters] */
} pvo_detector;

typedef struct {
    float score;
    int32_t filter;
    int32_t level, r, c;       /* position in the feature pyramid */
    int32_t l, t, rr, b;       /* rectangle in input-image pixels */
} pvo_det;

/* OpenMP threads used by the row-parallel loops (results do not depend on it) */
void pvo_set_threads(int n);
int pvo_get_max_threads(void);
int pvo_detector_levels(int h, int w, const pvo_detector* m);
/* raw candidates (score >= thresh), before NMS, in canonical order (sorted) */
int pvo_detect_raw(const uint8_t* rgb, int h, int w, int upsample, const pvo_detector* m,
                   double adjust, pvo_det* out, int cap);
int pvo_nms(const pvo_det* cands, int n, double iou, double covered, pvo_det* out, int cap);
int pvo_detect(const uint8_t* rgb, int h, int w, int upsample, const pvo_detector* m,
               double adjust, pvo_det* out, int cap);
/* debugging/stage access: build upsampled image + pyramid level l; returns dims */
int pvo_pyramid_level(const uint8_t* rgb, int h, int w, int upsample, int level, uint8_t* out,
                      int* oh, int* ow);
void pvo_score_level(const float* feat, int fh, int fw, const pvo_detector* m, int filter,
                     float* out /* [fh][fw], 0 on border */);

/* ---------------------------------------------------------------- ERT landmarks (pvo_ert.c) */
typedef struct {
    int n_cascades, n_trees, n_parts, n_pix, depth; /* 15, 500, 68, 500, 4 */
    const float* initial_shape;  /* [2*n_parts] x0,y0,x1,y1.. */
    const int32_t* anchor_idx;   /* [n_cascades][n_pix] */
    const float* deltas;         /* [n_cascades][n_pix][2] */
    const int32_t* split_idx1;   /* [n_cascades][n_trees][n_split] */
    const int32_t* split_idx2;
    const float* split_thresh;
    const float* leaves;         /* [n_cascades][n_trees][n_leaf][2*n_parts] */
} pvo_shape_model;
void pvo_landmarks(const uint8_t* rgb, int h, int w, const int32_t rect[4], const pvo_shape_model* m,
                   int32_t* pts /* [n_parts][2] */);

/* ---------------------------------------------------------------- face chip + ResNet (pvo_resnet.c) */
typedef struct {
    const float* mean_shape_xy; /* [51][2] template for landmarks 17..67 (dlib mean_face_shape_x/y) */
    int chip_size;              /* 150 */
    double chip_padding;        /* 0.25 */
    const float* blob;          /* all layer parameters, packed; see pvo_resnet.c for the walk order */
    size_t blob_len;
} pvo_embed_model;
void pvo_face_chip_details(const int32_t* pts68, const pvo_embed_model* m, pvo_chip_details* out);
void pvo_resnet_forward(const uint8_t* chip /*150x150x3*/, const pvo_embed_model* m, float* out128);
void pvo_embed(const uint8_t* rgb, int h, int w, const int32_t* pts68, const pvo_embed_model* m, float* out128);
size_t pvo_resnet_param_count(void);

/* ---------------------------------------------------------------- DSST tracker (pvo_dsst.c) */
typedef struct pvo_tracker pvo_tracker;
typedef struct {
    const double* mask64;    /* [64][64] radial cosine window (make_cosine_mask) */
    const double* mask_scale;/* [32] scale cosine window */
    const double* tw64;      /* [32][2] cos,sin(2*pi*k/64) */
    const double* tw32;      /* [16][2] cos,sin(2*pi*k/32) */
    double alpha_pow_m16;    /* scale_pyramid_alpha^(-16) */
    double ln_alpha;         /* ln(scale_pyramid_alpha) */
} pvo_dsst_tables;
pvo_tracker* pvo_tracker_new(const pvo_dsst_tables* t);
void pvo_tracker_free(pvo_tracker*);
void pvo_tracker_start(pvo_tracker*, const uint8_t* rgb, int h, int w, const double box[4]);
double pvo_tracker_update(pvo_tracker*, const uint8_t* rgb, int h, int w);
void pvo_tracker_position(const pvo_tracker*, double box[4]);
/* stage access for parity tests */
void pvo_tracker_debug_F(const pvo_tracker*, double* out /* [32][64][64][2] */);
void pvo_tracker_debug_state(const pvo_tracker*, double* A /*[32][64][64][2]*/, double* B /*[64][64]*/);
void pvo_fft64x64(double* data /* [64][64][2] in place */, const double* tw64, int inverse);
double pvo_det_exp(double x);

/* ---------------------------------------------------------------- association (pvo_assoc.c) */
void pvo_overlap_matrix(const double* a, int na, const double* b, int nb, double ratio, double* out);
/* munkres: square n x n cost matrix, returns column assigned to each row */
void pvo_munkres(const double* cost, int n, int32_t* row_to_col);

/* ---------------------------------------------------------------- clustering (pvo_cluster.c) */
/* T x T matrix of mean pairwise euclidean distances between the rows of track i and track j
 * (clustering.py:100-112); X float64 [N][128]; rows sorted by track, row_start[T+1]. */
void pvo_pair_mean_dist(const double* X, int N, int dim, const int32_t* row_start, int T, double* D);
/* average-linkage HAC from the track partition, stop at mean distance > threshold
 * (clustering.py:116-119,138-141 + pyannote.algorithms HAC [EXT]); label = smallest member index */
int pvo_hac(const double* D, const int32_t* sizes, int T, double threshold, int32_t* labels,
            double* merge_log /* optional [T-1][4]: a, b, dist, new size */);

#ifdef __cplusplus
}
#endif
/* ---- shot boundary detection (structure/shot.py:71-99), pvo_shot.c; PARITY UNPINNED (OpenCV internals restated) */
void pvo_shot_convert(const uint8_t* rgb, int ih, int iw, uint8_t* out /* [oh][ow] */, int oh, int ow);
void pvo_shot_tables(float* t /* 22: g[6] xg[6] xxg[6] ig11 ig03 ig33 ig55 */);
int pvo_farneback_small(const uint8_t* prev, const uint8_t* cur, int h, int w, const float* tables, float* flow /* [h][w][2] */);
int pvo_farneback_levels(int h, int w);   /* coarser pyramid levels OpenCV uses for this size (0: an image side below 64 pixels) */
int pvo_farneback(const uint8_t* prev, const uint8_t* cur, int h, int w, const float* tables, float* flow /* [h][w][2] */);
double pvo_shot_dfd_from_flow(const uint8_t* prev, const uint8_t* cur, int h, int w, const float* flow);
double pvo_shot_dfd(const uint8_t* prev, const uint8_t* cur, int h, int w, const float* tables);

#endif
