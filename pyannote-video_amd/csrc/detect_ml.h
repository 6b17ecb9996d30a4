// detect_ml.h -- what the multi-level detector kernels of detect.hip and screen.hip share: level descriptors, candidate records,
// the work items of the screening pass.
#pragma once
#include "pvf_internal.h"
#include "fhog_dev.h"

#define ML_MAX 32
struct LvDesc {
    int h, w, rb;                               // level image, row pitch in bytes (multiple of 64)
    int cells_nr, cells_nc, visible_nr, visible_nc;
    int fh, fw, hog_nr, hog_nc;                 // feature map (with its zero border) and the cells that carry features
    int fwp;                                    // cells per stored row of a plane group (fw + FEAT_PAD_COLS zero columns): see feat_at()
    int feat_bx, score_bx;
    int valid_score;
    int roll_nseg, roll_rows;                   // K3 v5: a column strip is walked in roll_nseg pieces of roll_rows output rows
    int strips, chunks, chunk_rows, fused_tasks;  // fused FHOG: 64-lane strips of 61 feature columns x chunks of chunk_rows feature rows
    long long img_off, img_stride;              // bytes
    long long feat_off, feat_stride;            // floats
};
struct MlStarts { int nl; int b0[ML_MAX + 1]; };

// Layout of a level's feature map (round 6): [row][plane group j = plane / 4][column][4 planes] -- a row of the map is 8 runs of fwp
// 16-byte pieces, one per plane group.  The FHOG kernel's lanes own one cell each, so store j of a wave writes the pieces of 61 neighbouring
// cells: ONE run of 976 bytes (with the cell-major [row][column][32 planes] of rounds 1-5 every store touched 64 lines with 16 bytes each:
// 3.0 of the kernel's 7.8 ms per 125 frames, tools/probes/store_pattern_probe.hip: 3.65 against 5.7 TB/s for the same bytes).  The readers
// fetch 16-byte pieces anyway (a slab of cells for the matrix cores); they address piece (row, j, column) instead of (row, column, j).
// A stored row is FEAT_PAD_COLS columns longer than the map and those columns hold zeros (written with the zero border): the screening
// pass reads up to 49 cells past the last column of a level's last strip and relies on finding zeros there.
#define FEAT_PAD_COLS 52
__host__ __device__ __forceinline__ size_t feat_at(int row, int j, int col, int fwp) { return (((size_t)row * 8 + j) * fwp + col) * 4; }   // in floats

struct ScoreParams { float thresh[8]; int n_filters; int level; int cap; };
struct CandRec { float score; int32_t filter, level, r, c; };

// ---- screen.hip: the f16 screening pass in front of the exact chain -------------------------------------------------------------
#define SCR_SG 4                                // 16-base groups per column strip: a strip is SCR_SG * 48 output columns wide
struct ScreenItem { int32_t lv, b, c_base, r_base, out_rows, ng; };      // one wave's walk: a strip of ng groups x out_rows output rows
// counters behind the per-frame candidate counts of a batch (d_counts + B): zeroed with them, copied back with them
enum { SCR_CURSOR = 0, SCR_FLAGGED = 1, SCR_VIOLATION = 2, SCR_CTL_INTS = 4 };

struct ScreenPlan {
    ScreenItem* d_items = nullptr;
    int n_items = 0;
    bool usable = false;                        // false: a level too large for the packed list entries => the dense kernel scores this plan
    ~ScreenPlan() { if (d_items) (void)hipFree(d_items); }
    ScreenPlan() = default;
    ScreenPlan(const ScreenPlan&) = delete;
    ScreenPlan& operator=(const ScreenPlan&) = delete;
};
// work items of one batch: every (frame, level, strip, piece of <= seg output rows), largest first
void screen_plan_build(ScreenPlan& sp, const std::vector<LvDesc>& lv, int B);
// screening + exact re-scoring of what it flags, on the detector stream; ctl = the SCR_CTL_INTS counters (zeroed by the caller)
void screen_launch(Ctx* c, const ScreenPlan& sp, const LvDesc* d_lv, int B, const float* feat, const ScoreParams& thr, int* d_counts,
                   CandRec* d_cands, int* ctl);
// the detector's weights as f16 B fragments + the error bound of the screening score per filter (ctx.hip calls it at model load)
void screen_prepare_model(DetectorModel& d, const float* w);
