/*
 * stvo_hip.h — C-ABI of the MI355X-native PL-StVO hot path (libstvo_hip.so, gfx950 only).
 *
 * The reference (/root/reference) has NO plugin / FFI interface: its boundary is the C++ class API
 * consumed by app/imagesStVO.cpp:86-124.  This header declares the `extern "C"` seams a maintainer
 * binds in place of the reference's inner functions (see INTEGRATION.md for the stubs); every
 * entry point cites the reference interface it replaces.  Plain pointers and sizes only.
 *
 *   - returns 0 (STVO_OK) or a negative STVO_ERR_* code; never throws, never aborts
 *   - there is NO CPU fallback: without a gfx950 device stvo_ctx_create fails with
 *     STVO_ERR_NO_DEVICE and nothing else can be called
 *   - host-buffer entry points copy in/out through pinned staging and synchronise before returning
 *   - *_dev entry points take DEVICE pointers, enqueue on the context's stream and do not
 *     synchronise (use stvo_ctx_synchronize / your own events)
 *   - descriptors: N x 32 bytes, contiguous; matrices row-major FP64; twists (t, w)
 */
#ifndef STVO_HIP_H
#define STVO_HIP_H

#include "stvo_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define STVO_ABI_VERSION 3
#define STVO_MAX_ROWS_LIMIT 65535 /* packed (distance << 16 | index) keys */
#define STVO_POSE_MAX_POINTS 2048 /* per frame pair: points owned per worker thread x worker threads of the pose kernel */
#define STVO_POSE_MAX_LINES 512

typedef struct stvo_ctx stvo_ctx;

/* ---- library / context ------------------------------------------------------------------- */
const char* stvo_backend_name(void); /* "hip-gfx950" */
int stvo_abi_version(void);
const char* stvo_error_string(int code);
/* Last HIP error text recorded on this context ("" if none). */
const char* stvo_ctx_last_error(const stvo_ctx* ctx);

/* One context per sequence / host thread (the reference's StereoFrameHandler is not re-entrant
 * either).  Owns device scratch sized for `max_rows` descriptors per set and
 * `max_batch` problems per launch, pinned staging buffers and (unless set_stream is used) a stream. */
int stvo_ctx_create(int device_id, int max_rows, int max_batch, stvo_ctx** out);
int stvo_ctx_destroy(stvo_ctx* ctx);
/* Borrow an existing hipStream_t (e.g. the caller's / torch's current stream); NULL = default. */
int stvo_ctx_set_stream(stvo_ctx* ctx, void* hip_stream);
int stvo_ctx_synchronize(stvo_ctx* ctx);
/* Throughput option for the batched path.  enable = 1: stvo_track_batched_dev enqueues the pose kernel
 * on a second (context-owned) stream behind an event, so that the NEXT call's matching kernels run
 * concurrently with this call's latency-bound pose kernel (worth ~7 % with the VALU matcher, which is then
 * launched with an occupancy cap, ~1-2 % with the matrix-core matcher).  Results of a call are complete after stvo_ctx_synchronize (or a
 * device synchronise); hazards between consecutive calls on the same buffers are handled with events.
 * enable = 0 (default): strict stream order on the context's stream. */
int stvo_ctx_set_overlap(stvo_ctx* ctx, int enable);

/* ---- matching: host-buffer seams ------------------------------------------------------------ */

/* Replaces  int StVO::match(const cv::Mat&, const cv::Mat&, float nnr, std::vector<int>&)
 * (include/matching.h:52, src/matching.cpp:63-91; with mutual == 0: matchNNR, :41-61).
 * m12[i] = matched row of d2 or -1;  *n_matches = return value of the reference function.
 * n2 < 2 yields no matches (the reference reads out of bounds there, src/matching.cpp:54). */
int stvo_match_nnr_mutual(stvo_ctx* ctx, const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int mutual,
                          int32_t* m12, int32_t* n_matches);

/* Replaces  int StVO::matchGrid(const std::vector<point_2d>&, const cv::Mat&, const GridStructure&,
 *                               const cv::Mat&, const GridWindow&, std::vector<int>&)
 * (include/matching.h:57, src/matching.cpp:111-177).  The GridStructure (64x48 std::list buckets,
 * src/gridStructure.cpp:43-83) is passed as CSR: cell c = y*64 + x owns
 * cell_items[cell_start[c] .. cell_start[c+1]).  `ratio` = Config::minRatio12P() (double test, :160),
 * `mutual` = Config::bestLRMatches(). */
int stvo_match_grid_points(stvo_ctx* ctx, const int32_t* cell_xy1 /*[n1][2]*/, const uint8_t* d1, int n1,
                           const int32_t* cell_start /*[3073]*/, const int32_t* cell_items, const uint8_t* d2, int n2,
                           const stvo_grid_window* w, double ratio, int mutual, int32_t* m12, int32_t* n_matches);

/* Replaces the line overload of StVO::matchGrid (include/matching.h:60, src/matching.cpp:179-258).
 * cell_xy1 = [n1][4] (sx, sy, ex, ey) integer cells of the left end points; dir2 = [n2][2] unit
 * directions of the right lines in grid space (src/stereoFrame.cpp:331-333). */
int stvo_match_grid_lines(stvo_ctx* ctx, const int32_t* cell_xy1 /*[n1][4]*/, const uint8_t* d1, int n1,
                          const int32_t* cell_start /*[3073]*/, const int32_t* cell_items, const uint8_t* d2, int n2,
                          const double* dir2 /*[n2][2]*/, const stvo_grid_window* w, double ratio, double line_sim_th,
                          int mutual, int32_t* m12, int32_t* n_matches);

/* ---- optimizer: host-buffer seams ----------------------------------------------------------- */

/* Replaces the private  void StereoFrameHandler::optimizeFunctions(Matrix4d, Matrix6d&, Vector6d&, double&)
 * and optimizeFunctionsRobust (include/stereoFrameHandler.h:97-98, src/stereoFrameHandler.cpp:549-962):
 * one evaluation of H (6x6), g (6), e at pose T over the inlier records. *n_used = N_p + N_l. */
int stvo_normal_eq(stvo_ctx* ctx, const double T[16], const stvo_cam* cam, const stvo_opt_params* params,
                   const stvo_matched* m, int robust, double H[36], double g[6], double* e, int32_t* n_used);

/* Replaces  void StereoFrameHandler::optimizePose()  (include/stereoFrameHandler.h:54,
 * src/stereoFrameHandler.cpp:307-392) minus the Tfw composition (:377-378, done by the caller):
 * GN / robust GN / LM, removeOutliers (:988-1067), isGoodSolution (:292-305) and the commit rule.
 * init_T = the DT chosen at :317-326.  m->inlier_* are updated in place. */
int stvo_optimize_pose(stvo_ctx* ctx, const double init_T[16], const stvo_cam* cam, const stvo_opt_params* params,
                       stvo_matched* m, stvo_pose_result* out);

/* ---- batched, device-resident path (throughput mode; B independent frame pairs) --------------- */

/* All pointers are DEVICE pointers.  Per-frame arrays are strided by max_pts / max_lines rows.
 * Replaces, per frame pair, f2fTracking() + optimizePose() (src/stereoFrameHandler.cpp:106-180,
 * 307-392): brute-force mutual NNR matching of prev->pdesc_l vs curr->pdesc_l (and ldesc_l),
 * gathering of the matched records, and the full pose optimisation. */
typedef struct stvo_track_batch_dev {
    int32_t B, max_pts, max_lines, reserved;
    /* prev frame: stereo points / lines that survived stereo association */
    const int32_t* n_prev_pts;   /* [B] */
    const uint8_t* prev_pdesc;   /* [B][max_pts][32] */
    const double* prev_P;        /* [B][max_pts][3] */
    const double* prev_sigma2p;  /* [B][max_pts] */
    const int32_t* n_curr_pts;   /* [B] */
    const uint8_t* curr_pdesc;   /* [B][max_pts][32] */
    const double* curr_pl;       /* [B][max_pts][2]  -> pl_obs of the matched prev point */
    const int32_t* n_prev_lines; /* [B]  (may be NULL when max_lines == 0) */
    const uint8_t* prev_ldesc;   /* [B][max_lines][32] */
    const double* prev_sP;       /* [B][max_lines][3] */
    const double* prev_eP;       /* [B][max_lines][3] */
    const double* prev_spl;      /* [B][max_lines][2] */
    const double* prev_epl;      /* [B][max_lines][2] */
    const double* prev_sigma2l;  /* [B][max_lines]  (after safeCopy's re-scaling) */
    const int32_t* n_curr_lines; /* [B] */
    const uint8_t* curr_ldesc;   /* [B][max_lines][32] */
    const double* curr_le;       /* [B][max_lines][3] -> le_obs of the matched prev line */
    const double* init_T;        /* [B][16] or NULL = identity */
    /* outputs */
    int32_t* m12_pts;            /* [B][max_pts]   */
    int32_t* m12_lines;          /* [B][max_lines] */
    int32_t* inlier_pts;         /* [B][max_pts]   -1 unmatched, 0 outlier, 1 inlier */
    int32_t* inlier_lines;       /* [B][max_lines] */
    stvo_pose_result* results;   /* [B] */
} stvo_track_batch_dev;

int stvo_track_batched_dev(stvo_ctx* ctx, const stvo_track_batch_dev* batch, const stvo_cam* cam,
                           const stvo_opt_params* params, float nnr_points, float nnr_lines, int mutual);

/* The matching stage alone (forward scan + mutual check) on device-resident descriptor sets: m12[b][i]. */
int stvo_match_nnr_mutual_batched_dev(stvo_ctx* ctx, int B, int row_stride, const uint8_t* d1, const int32_t* n1,
                                      const uint8_t* d2, const int32_t* n2, float nnr, int mutual, int32_t* m12);

/* The optimizer stage alone on device-resident, already associated records (m12 = identity). */
int stvo_optimize_pose_batched_dev(stvo_ctx* ctx, const stvo_track_batch_dev* batch, const stvo_cam* cam,
                                   const stvo_opt_params* params);

/* ---- device-resident per-frame pipeline ------------------------------------ */

/* B independent stereo sequences advancing in lock-step; all per-frame state (stereo point / line sets of the
 * previous frame, grids, matches) lives in HBM.  Replaces, per pushed frame and per sequence,
 * StereoFrameHandler::initialize / insertStereoPair + optimizePose + updateFrame minus feature detection
 * (src/stereoFrameHandler.cpp:35-60, 307-392, 89-100): stereo association on the 64x48 grid
 * (src/stereoFrame.cpp:120-173, 309-415), f2f tracking, pose optimisation.  The Tfw composition (:377-378) and the
 * adaptive FAST threshold (:66-86, a front-end knob) stay with the caller.  init_T is the identity
 * (use_motion_model = false, as in every shipped configuration) unless stvo_seq_set_motion_model turns the motion model on. */
typedef struct stvo_seq stvo_seq;
int stvo_seq_create(stvo_ctx* ctx, int B, int max_keypoints, int max_keylines, int img_cols, int img_rows,
                    const stvo_cam* cam, const stvo_match_params* mp, const stvo_opt_params* op, stvo_seq** out);
/* As stvo_seq_create with one calibration and image size PER SEQUENCE: cams[B], img_cols[B], img_rows[B] (host arrays).
 * BASELINE configs[4] runs KITTI sequences 00-07 side by side and the reference builds one PinholeStereoCamera per
 * dataset (app/imagesStVO.cpp:66): config/dataset_params/kitti00-02.yaml, kitti03.yaml and kitti04-10.yaml differ in
 * focal length, principal point, baseline and image size (1241x376, 1242x375, 1226x370), i.e. in the grid scale
 * 64 / cols, 48 / rows of src/stereoFrame.cpp:47-48 as well. */
int stvo_seq_create_multi(stvo_ctx* ctx, int B, int max_keypoints, int max_keylines, const int32_t* img_cols,
                          const int32_t* img_rows, const stvo_cam* cams, const stvo_match_params* mp,
                          const stvo_opt_params* op, stvo_seq** out);
int stvo_seq_destroy(stvo_seq* seq);
/* Config::useMotionModel() for every sequence of the pipeline (src/stereoFrameHandler.cpp:317-324): the initial DT of a frame pair is
 * prev_frame->DT (the increment COMMITTED for the previous pair, i.e. after :374's inverse) unless !isGoodSolution(prev_frame->DT,
 * prev_frame->DT_cov, prev_frame->err_norm), then the identity.  The decision is taken on the device by the previous step's commit —
 * the result never leaves HBM.  Enabling (again) restarts every sequence from prev_frame->DT = I (:45); synchronises. */
int stvo_seq_set_motion_model(stvo_seq* seq, int enable);
/* Number of raw frame slots (2 .. STVO_SEQ_MAX_SLOTS; 2 after create).  A throughput caller uploads several consecutive
 * frames of every sequence once (stvo_seq_upload) and rotates stvo_seq_step_dev through the slots. */
#define STVO_SEQ_MAX_SLOTS 16
int stvo_seq_set_slots(stvo_seq* seq, int n_slots);
/* One upload, one synchronisation, one small download per frame.  results: [B] (zeroed for the first frame, which
 * only builds the stereo sets); counts: optional [B][4] = stereo points, stereo lines, matched points, matched
 * lines of this frame. */
int stvo_seq_push(stvo_seq* seq, const stvo_frame_features* frame, stvo_pose_result* results, int32_t* counts);
/* The three halves of stvo_seq_push, for callers that keep frames resident in HBM (throughput mode): upload one
 * frame's features into device slot 0 / 1 (asynchronous), run the pipeline on a resident slot (asynchronous, no
 * host transfer), fetch the results of the last step (synchronises). */
int stvo_seq_upload(stvo_seq* seq, int slot, const stvo_frame_features* frame);
/* stvo_seq_upload for features that already live in DEVICE memory — every pointer of `frame`, the count arrays included; oct_l /
 * oct_ll may be NULL (octave 0, e.g. the one-level ORB front-end below).  Asynchronous on the context's stream; counts are
 * clamped to the capacities on the device.  Together with stvo_orb_detect_dev: images in, poses out, nothing through the host
 * (replaces detectStereoPoints + matchStereoPoints + f2fTracking + optimizePose, src/stereoFrame.cpp:88-173,
 * src/stereoFrameHandler.cpp:106-392). */
int stvo_seq_upload_dev(stvo_seq* seq, int slot, const stvo_frame_features* frame);
int stvo_seq_step_dev(stvo_seq* seq, int slot);
int stvo_seq_read(stvo_seq* seq, stvo_pose_result* results, int32_t* counts);

/* By-products for callers that keep the reference's HOST-side feature lists (the StereoFrameHandler mirror):
 * with fetch enabled every step also copies to pinned memory (a) right after the f2f stage — while the pose
 * kernel still runs — the raw stereo matches of the new frame (left key-point / key-line i -> right index or -1,
 * what matchGrid returns at stereoFrame.cpp:145,344, [B][K] / [B][M] with K, M = the capacities rounded up to 64)
 * and the f2f matches (prev stereo feature k -> curr stereo feature or -1, what StVO::match returns at
 * stereoFrameHandler.cpp:141,164), and (b) after the pose kernel the inlier flags of the prev stereo features
 * (-1 unmatched, 0 outlier, 1 inlier; removeOutliers :988-1067).  fetch_matches waits for (a) only; the pointers
 * stay valid until the next step.  fetch_inliers synchronises the stream. */
int stvo_seq_enable_fetch(stvo_seq* seq, int enable);
int stvo_seq_fetch_matches(stvo_seq* seq, const int32_t** m12_stereo_pts, const int32_t** m12_stereo_lines,
                           const int32_t** m12_pts, const int32_t** m12_lines);
int stvo_seq_fetch_inliers(stvo_seq* seq, const int32_t** inl_pts, const int32_t** inl_lines);
/* Row strides of the arrays above. */
int stvo_seq_strides(const stvo_seq* seq, int32_t* stride_pts, int32_t* stride_lines);

/* Live per-stage timing of the pipeline (bench.py): with enable = 1 every stvo_seq_step_dev brackets, with hipEvents on
 * the stream the kernels run on, stage 0 = the whole stereo point stage (cells + grid matcher + tail), 1 = the two
 * grid_scan passes of the point grid matcher alone, 2 = the forward top-2 scan of the point f2f match (K1m), 3 = plan +
 * reverse scans of its mutual check, 4 = the pose kernel.  get synchronises, returns the average milliseconds per stage over
 * the steps measured since the last get / set (n_steps = the largest number of samples any stage has) and resets.
 * enable = 2 ("light"): pairs around stages 1, 2 and 4 only — the three big kernels of the point stream — and the step is otherwise
 * enqueued exactly as an untimed one (key-line stream unmarked, the next frame's grid built ahead on it), so the kernels keep the
 * neighbours they have in the timed region; stages 0 and 3 report 0. */
#define STVO_SEQ_NSTAGE 5
int stvo_seq_set_stage_timing(stvo_seq* seq, int enable);
int stvo_seq_get_stage_timing(stvo_seq* seq, float avg_ms[STVO_SEQ_NSTAGE], int32_t* n_steps);

/* TEST HOOK: the grid structures the last step built ON THE DEVICE for sequence b (lines = 0: key-points, 1: key-lines):
 * cell_start[3073] / cell_items (cell c = y * 64 + x owns cell_items[cell_start[c] .. cell_start[c+1]); the order inside a
 * cell is unspecified) = GridStructure after src/stereoFrame.cpp:135-139 / :325-338 (lines: every LineIterator cell);
 * cells_left [n_left][2 | 4] = the integer cells of the left features (:129-132, :318-322); cand_off[n_left + 1] / cand =
 * the candidate set GridStructure::get (src/gridStructure.cpp:65-76) yields for every left feature with the stereo window
 * (:141-143, :340-342; lines: the union over both end points, src/matching.cpp:213-215), ascending right ids. */
int stvo_seq_debug_grid(stvo_seq* seq, int b, int lines, int32_t* cell_start, int32_t* cell_items, int32_t cap_items,
                        int32_t* cells_left, int32_t* cand_off, int32_t* cand, int32_t cap_cand, int32_t* n_left);

/* ---- ORB point front-end ------------------------------------------------------------------ */

/* Replaces  cv::ORB::create(nfeatures, scaleFactor, nlevels, ...)->detectAndCompute(img, Mat(), points, pdesc, false)  as
 * called by StereoFrame::detectPointFeatures (src/stereoFrame.cpp:104-118) for B images of cols x rows bytes at once: the image
 * pyramid (8-bit bilinear resize level by level), per level FAST-9/16 with non-maximum suppression, border filter,
 * retainBest(the level's share of nfeatures) on the FAST response (ties at the cut are kept), intensity-centroid orientation,
 * 7x7 Gaussian blur and the 256-bit rotated BRIEF descriptor; key-points of all levels in level order with their octave and
 * coordinates scaled back to the full image.  Within a level key-points are emitted in row-major order (OpenCV leaves the
 * order to std::nth_element).  When more key-points qualify than max_keypoints holds, the first max_keypoints of that order
 * come back (deterministically) and the uncapped count is reported in n_total.  max_keypoints may not exceed 4096
 * (STVO_ERR_CAPACITY).  OpenCV is third-party code that is not part of the reference tree: the semantics are pinned to
 * oracle/stvo_orb_oracle.c only (parity unpinned). */
typedef struct stvo_orb stvo_orb;
int stvo_orb_create(stvo_ctx* ctx, int B, int cols, int rows, int max_keypoints, const stvo_orb_params* prm, stvo_orb** out);
int stvo_orb_destroy(stvo_orb* orb);
/* The 256 x 4 test pattern (x0, y0, x1, y1 per bit, |coordinate| <= 13).  The default is a seeded table; OpenCV's learned
 * table (bit_pattern_31_ of features2d/src/orb.cpp) is data this repository does not hold — pass it here to reproduce it. */
int stvo_orb_set_pattern(stvo_orb* orb, const int8_t* pattern /*[1024]*/);
/* The FAST threshold of the following calls: StereoFrameHandler::updateFrame adapts orb_fast_th frame by frame
 * (src/stereoFrameHandler.cpp:66-86) and passes it to detectStereoFeatures (:56 -> src/stereoFrame.cpp:59,104-118). */
int stvo_orb_set_fast_threshold(stvo_orb* orb, int fast_threshold /* 1 .. 254 */);
int stvo_orb_get_pattern(const stvo_orb* orb, int8_t* pattern /*[1024]*/);
/* Host buffers in / out, synchronises.  images [B][rows][cols]; kp_xy [B][max_keypoints][2] = cv::KeyPoint::pt; response =
 * cv::KeyPoint::response (FAST score); angle in degrees = cv::KeyPoint::angle; desc [B][max_keypoints][32]; n_kp [B]. */
int stvo_orb_detect(stvo_orb* orb, const uint8_t* images, float* kp_xy, float* response, float* angle, uint8_t* desc,
                    int32_t* n_kp);
/* The same with DEVICE pointers, enqueued on the context's stream (no synchronisation). */
int stvo_orb_detect_dev(stvo_orb* orb, const uint8_t* images, float* kp_xy, float* response, float* angle, uint8_t* desc,
                        int32_t* n_kp);
/* The full forms: octave [B][max_keypoints] = cv::KeyPoint::octave (the pyramid level, what PointFeature's sigma2 is made of,
 * src/stereoFeatures.cpp:41-47) and n_total [B] = the number of key-points that qualified before the max_keypoints cap
 * (n_total > n_kp: the output was truncated).  Either may be NULL. */
int stvo_orb_detect_levels(stvo_orb* orb, const uint8_t* images, float* kp_xy, float* response, float* angle, int32_t* octave,
                           uint8_t* desc, int32_t* n_kp, int32_t* n_total);
int stvo_orb_detect_levels_dev(stvo_orb* orb, const uint8_t* images, float* kp_xy, float* response, float* angle, int32_t* octave,
                               uint8_t* desc, int32_t* n_kp, int32_t* n_total);

/* ---- LBD line descriptor (first half) ------------------------------------------------------- */

/* Replaces  BinaryDescriptor::createBinaryDescriptor()->compute(img, lines, ldesc)  as called by StereoFrame::detectLineFeatures
 * (src/stereoFrame.cpp:213,243,303) for B images of cols x rows bytes with up to max_keylines octave-0 key-lines each:
 * GaussianBlur(5 x 5, sigma 1), Sobel 3 x 3, the 9-band statistics of the 63-row line support region in float and in the source's
 * order of operations, both normalisations, and the 32-byte binary form (3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:
 * 350-412, 539-687, 1026-1340 — source the reference holds; semantics pinned to oracle/stvo_lbd_oracle.c, which cites it line by
 * line; parity unpinned because that file needs OpenCV to build).  The key-lines themselves come from the caller: the LSD / FLD
 * detectors are not built. */
typedef struct stvo_lbd stvo_lbd;
int stvo_lbd_create(stvo_ctx* ctx, int B, int cols, int rows, int max_keylines, stvo_lbd** out);
int stvo_lbd_destroy(stvo_lbd* lbd);
/* Host buffers in / out, synchronises.  images [B][rows][cols]; lines [B][max_keylines]; n_lines [B]; desc [B][max_keylines][32] =
 * the rows of ldesc; desc_float (may be NULL) [B][max_keylines][72] = the float descriptor (returnFloatDescr). */
int stvo_lbd_compute(stvo_lbd* lbd, const uint8_t* images, const stvo_keyline* lines, const int32_t* n_lines, uint8_t* desc,
                     float* desc_float);
/* The same with DEVICE pointers, enqueued on the context's stream (no synchronisation). */
int stvo_lbd_compute_dev(stvo_lbd* lbd, const uint8_t* images, const stvo_keyline* lines, const int32_t* n_lines, uint8_t* desc,
                         float* desc_float);

/* ---- LSD line detector (second half) -------------------------------------------------------- */

/* Replaces  lsd->detect(img, lines, Config::lsdScale(), 1, opts)  + the top-N cut by response of StereoFrame::detectLineFeatures
 * (src/stereoFrame.cpp:219-240; LSDDetectorC::detectImpl, 3rdparty/line_descriptor/src/LSDDetector_custom.cpp:227-325, one
 * octave) for B images of cols x rows bytes: cv::LineSegmentDetector — third-party code the reference does not hold — as restated
 * in oracle/stvo_lsd_oracle.c from the published algorithm (parity unpinned), lsd_refine = 0 (LSD_REFINE_NONE: what every
 * shipped configuration uses) or 1 (LSD_REFINE_STD: a region too sparse for its rectangle is given back, grown again under a tolerance
 * from the angles near its seed and cut back by radius — one wavefront per image for every batch size, since flags then also turn
 * off); 2 (LSD_REFINE_ADV: rect_improve + the NFA test): STVO_ERR_UNSUPPORTED.  Images up to 2^20 pixels after scaling.  Blur, resize, gradient /
 * level-line angles and the pseudo-ordering are data-parallel kernels; region growing is inherently sequential per image (a
 * pixel joins a region depending on the running region angle and on what every earlier region took) and runs as ONE wavefront
 * per image, the images of the batch side by side. */
typedef struct stvo_lsd_params {
    int32_t refine;        /* Config::lsdRefine()      0 (or 1) */
    int32_t n_bins;        /* Config::lsdNBins()       1024 (1 .. 2048; more: STVO_ERR_UNSUPPORTED) */
    double scale;          /* Config::lsdScale()       1.2 */
    double sigma_scale;    /* Config::lsdSigmaScale()  0.6 */
    double quant;          /* Config::lsdQuant()       2.0 */
    double ang_th;         /* Config::lsdAngTh()       22.5 */
    double log_eps;        /* Config::lsdLogEps()      (unused: it belongs to refine 2) */
    double density_th;     /* Config::lsdDensityTh()   0.6 (read with refine 1) */
    double min_length;     /* LSDOptions::min_length = min_line_length x min(cols, rows) (stereoFrame.cpp:78) */
    int32_t nfeatures;     /* Config::lsdNFeatures()   300, 0: keep all */
    int32_t reserved;
} stvo_lsd_params;
typedef struct stvo_lsd stvo_lsd;
int stvo_lsd_create(stvo_ctx* ctx, int B, int cols, int rows, int max_keylines, const stvo_lsd_params* prm, stvo_lsd** out);
int stvo_lsd_destroy(stvo_lsd* lsd);
/* Host buffers in / out, synchronises.  images [B][rows][cols]; lines [B][max_keylines] (what stvo_lbd_compute consumes: in-octave
 * end points, angle, LineIterator count), response (may be NULL) [B][max_keylines] = KeyLine::response, n_lines [B]; lines come
 * in detection order, or by descending response when the top-N cut applied — also when max_keylines is what cuts (nfeatures 0 or
 * above the capacity): the strongest max_keylines lines are kept, and stvo_lsd_counts tells that a cut happened. */
int stvo_lsd_detect(stvo_lsd* lsd, const uint8_t* images, stvo_keyline* lines, float* response, int32_t* n_lines);
/* Counts of the last detection before its cuts, host arrays [B] (either may be NULL), synchronises: n_segments = segments the
 * detector core found (the first 8192 are ranked), n_passing = those longer than min_length. */
int stvo_lsd_counts(stvo_lsd* lsd, int32_t* n_segments, int32_t* n_passing);
/* The same with DEVICE pointers, enqueued on the context's stream (no synchronisation). */
int stvo_lsd_detect_dev(stvo_lsd* lsd, const uint8_t* images, stvo_keyline* lines, float* response, int32_t* n_lines);
/* Device helper between stvo_lsd_detect_dev / stvo_lbd_compute_dev and stvo_seq_upload_dev: the end points (sx, sy, ex, ey) of
 * the key-line records [B][stride] as the float rows [B][stride][4] that stvo_frame_features::kl_l / kl_r take (device pointers,
 * enqueued on the context's stream; rows beyond n_lines[b] are zeroed). */
int stvo_keylines_xy_dev(stvo_ctx* ctx, int B, int stride, const stvo_keyline* lines, const int32_t* n_lines, float* kl_xy);
/* test hook: the raw segments of the detector core (cv::LineSegmentDetector::detect), host buffers, synchronises:
 * segments [B][cap][4], n_segments [B] (all found; at most cap stored) */
int stvo_lsd_segments(stvo_lsd* lsd, const uint8_t* images, float* segments, int cap, int32_t* n_segments);

/* ---- measurement helpers --------------------------------------------------------------------- */
/* Times `iters` launches of the named kernel stage on the context's stream with hipEvents and
 * returns the average milliseconds per launch (used by bench.py for the roofline line).
 * stage: 0 = the forward top-2 scan (K1m hamming_knn2_mfma, or K1 hamming_knn2 for contexts beyond 8192 rows /
 * STVO_KNN_MFMA=0), 1 = pose kernel, 3 = the reverse-check scans (K1m's two sparse scans, or K1v hamming_verify) on
 * the column claims and lists left by the last stvo_track_batched_dev call. */
int stvo_time_stage_dev(stvo_ctx* ctx, const stvo_track_batch_dev* batch, const stvo_cam* cam,
                        const stvo_opt_params* params, float nnr, int stage, int iters, float* avg_ms);

/* Live timing of the dominant kernel INSIDE the batched path: with enable = 1 every stvo_track_batched_dev
 * call brackets its forward top-2 scan and its reverse (mutual) check — planning kernel + scans — on
 * the point descriptors with hipEvents on the stream they are launched on.  get_kernel_timing synchronises,
 * returns the average duration of each and the number of calls measured since the last get / set, and
 * resets the pool. */
int stvo_ctx_set_kernel_timing(stvo_ctx* ctx, int enable);
int stvo_ctx_get_kernel_timing(stvo_ctx* ctx, float* avg_ms_forward, float* avg_ms_reverse, int32_t* n_calls);

/* Number of right-hand (curr) rows the LAST batched mutual match had to verify, per frame pair (only columns
 * claimed by some row's accepted forward match are examined). */
int stvo_last_reverse_counts(stvo_ctx* ctx, int B, int32_t* counts);
/* The plan of the LAST mutual match's reverse check on the matrix cores (B = the batch size of that call; 1 for
 * stvo_match_nnr_mutual), plan = [5][B]: claimed columns; LIGHT columns (verified against the few rows whose second-best
 * forward distance is within the cut); HEAVY columns (verified against every row); |S| = number of such rows; the cut tau
 * (-1: every column heavy).  Diagnostics for tests and bench.py; meaningless after a match that took the VALU kernels. */
int stvo_last_reverse_plan(stvo_ctx* ctx, int B, int32_t* plan);

/* Integer-VALU micro-benchmark (xor + popcount-accumulate chains, no memory traffic): measured
 * 32-bit lane-ops/s of this device, the empirical roof K1 is priced against. */
int stvo_valu_peak_probe(stvo_ctx* ctx, double* lane_ops_per_s);

/* Developer switches (csrc/debug_switches.h lists every STVO_* environment variable the library knows): they are parsed once,
 * on first use; this re-reads the environment so that one test process can drive several kernel variants. */
void stvo_debug_reparse_env(void);

#ifdef __cplusplus
}
#endif
#endif /* STVO_HIP_H */
