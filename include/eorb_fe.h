/*
 * eorb_fe.h -- C ABI of the MI355X-native event-frame front end (libeorb_fe.so).
 *
 * Drop-in boundary for the hot path of m-dayani/EORB_SLAM: event->image accumulation, ORB
 * extraction and 256-bit Hamming matching.  The reference has no FFI for this path: it is reached
 * through three C++ seams, and every entry point below names the seam (file:line in the
 * reference repository) it replaces.  INTEGRATION.md shows the adapter a maintainer adds on the
 * reference side.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every call returns 0 on success, <0 on error (EORB_E_*); never throws, never aborts.
 *   - an eorb_ctx owns device workspaces and runs on ONE HIP stream; it is single-threaded: one
 *     thread uses it at a time.  The reference also calls ev2im_gauss from 4 transient threads per
 *     motion-compensated image (src/Event/EvImBuilder.cpp:1165-1193): let those BORROW warm contexts
 *     from a pool instead of creating one per thread (eorb_host::ContextPool).
 *   - "host" entry points take host pointers, copy in/out and synchronise before returning.
 *   - "_dev" entry points take DEVICE pointers (HBM resident), enqueue on the ctx stream and do
 *     not synchronise: this is the throughput path (batches of slices).
 *   - results are bit-identical to the strict-IEEE CPU restatement of the reference (oracle/).
 */
#ifndef EORB_FE_H
#define EORB_FE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EORB_OK            0
#define EORB_E_EMPTY      -1   /* empty image: ORBextractor::operator() returns -1 (ORBextractor.cc:1096) */
#define EORB_E_CONFIG     -2   /* configuration the reference cannot run (division by zero) */
#define EORB_E_CAPACITY   -3   /* caller buffer / configured capacity too small */
#define EORB_E_ARG        -4   /* bad argument */
#define EORB_E_HIP        -5   /* HIP runtime error (see eorb_last_error) */
#define EORB_E_NOTCONF    -6   /* call needs a prior eorb_*_configure */

typedef struct eorb_ctx eorb_ctx;

/* include/Event/EventData.h:36-58 : struct EventData {double ts; float x; float y; bool p} (24 B) */
typedef struct {
    double  ts;
    float   x, y;
    uint8_t p;
    uint8_t pad_[7];
} eorb_event;

/* HBM-resident compact event record (16 B = the algorithmic bytes/event of SURVEY §8(d)):
 * x, y as the loader produced them; t = timestamp with the polarity packed in its sign bit
 * (t >= 0 always: sign bit SET means p == false). */
typedef struct {
    float  x, y;
    double t;
} eorb_event16;

/* one event as the sensor / dataset delivers it (src/Event/EventLoader.cpp:80-92 "ts x y p"): integer pixel, polarity,
 * timestamp.  16 B, the same HBM footprint as eorb_event16. */
typedef struct {
    uint16_t x, y;
    uint32_t p;         /* 0 = negative, otherwise positive */
    double   t;
} eorb_raw_event;

/* the same event as a 4-byte wire record for the event IMAGES (they never read the time stamp: ev2im / ev2im_gauss use x, y and the
 * polarity, src/Event/EventConversion.cc:173-269): x | p << 15 | y << 16, sensor sizes up to 32767 x 65535.  A quarter of the bytes
 * on the host -> HBM link (bench.py --stream). */
typedef uint32_t eorb_raw_event4;
/* the 2-byte wire record: the sensor pixel's linear index y * LW + x in the maps of eorb_set_undistort_maps (0xffff = no event); for
 * polarity-free images on sensors of at most 65 535 pixels, which read nothing else of an event */
typedef uint16_t eorb_raw_event2;

/* cv::KeyPoint (28 B): what ORBextractor::operator() fills (_keypoints) */
typedef struct {
    float   x, y;
    float   size;
    float   angle;
    float   response;
    int32_t octave;
    int32_t class_id;
} eorb_keypoint;

/* include/ORBextractor.h:33-47 : struct ORBxParams */
typedef struct {
    int   nfeatures;
    float scaleFactor;
    int   nlevels;
    int   iniThFAST;
    int   minThFAST;
    int   edgeTh;       /* Features.imMargin; <0 = adaptive rule of ORBextractor.cc:481-488 */
    int   imWidth;      /* used only by the adaptive rule */
} eorb_orb_params;

/* Frame image bounds + grid pitch: Frame::mnMinX.., mfGridElementWidthInv (Frame.cc:362-363, 855-866) */
typedef struct {
    float minX, minY, maxX, maxY;
    float invW, invH;
} eorb_grid_bounds;

/* ---- context --------------------------------------------------------------------------------- */
/* hip_stream: a hipStream_t to launch on (e.g. torch's current stream), or NULL for a private one */
int         eorb_create(int device, void* hip_stream, eorb_ctx** out);
void        eorb_destroy(eorb_ctx* ctx);
/* waits for the ctx stream; also reports (once, then clears) the sticky status of the asynchronous *_dev paths:
 * EORB_E_CAPACITY when a kernel of an earlier batched call overflowed an internal capacity (its keypoints are truncated).
 * The host-buffer entry points report the same condition from the call itself (eorb_orb_extract). */
int         eorb_sync(eorb_ctx* ctx);
/* test hooks, not part of the reference's interface: "octree_pool_shrink" (n > 0: shrink the octree node pool by n at the
 * next configure, to force the overflow path), "octree_force_global" (1: keep the whole octree working set in global memory), "orb_three_launches" (1: orientation, descriptors and output order as the three kernels of a call with a lapping area), "win_lds_entries" (window matchers: entries of a pair that the second phase stages in LDS; default: as many as fit), "win_list_cap" / "win_pool_cap" (window matchers:
 * candidate list capacity per query / pool per pair, to force the full-scan path), "gather_form" (raw events with a Gaussian
 * stamp: 0 = choose the gather kernel by the batch's shape, 1 = the pipelined workgroup per tile, 2 = the wave per tile, 3 = no
 * binning, every tile's wave reads all events of its slice (calls with at most 4 slices), 4 = two-byte slot lists whatever the
 * batch's shape (falls back to 1 where they do not apply: polarity, sigma > 4/3, a tile with more than 254 slots)), "dedupe_min_events" (float events:
 * number of events per call from which their distinct positions are tabulated, default 2^20) */
int         eorb_debug_option(eorb_ctx* ctx, const char* name, int value);
/* test hook: counters a test can read to see which path served its calls.  "slot_calls": accumulation calls that took the slot
 * lists (gather_form 0 on dense batches, or 4); "slot_flags": sticky device flags of that path (0 = fine; synchronises); "slot_hot_items": lists the last such call handed to
 * the register-row kernel (synchronises); "slot_rank_ok": 1 when the scatter takes its ranks from LDS atomics.
 * Returns the value, or -1 for an unknown name. */
long long   eorb_debug_counter(eorb_ctx* ctx, const char* name);
const char* eorb_last_error(eorb_ctx* ctx);
const char* eorb_version(void);
/* per-kernel HIP-event timing on the ctx stream (off by default; used by bench.py) */
int         eorb_prof_enable(eorb_ctx* ctx, int on);
int         eorb_prof_reset(eorb_ctx* ctx);
/* names: comma-separated scope names to time, NULL or "" = all (two event records per scope and call are a visible share of a
 * short step: bench.py times only the accumulation scopes inside its timed steps) */
int         eorb_prof_only(eorb_ctx* ctx, const char* names);
int         eorb_prof_count(eorb_ctx* ctx);
int         eorb_prof_get(eorb_ctx* ctx, int i, const char** name, double* total_ms, int64_t* launches);

/* ---- event accumulation (host buffers) -------------------------------------------------------- */
/* replaces EvImConverter::ev2im, src/Event/EventConversion.cc:173-212 (include/Event/EventConversion.h:52).
 * out_f32 (W*H, optional) = accumulated CV_32FC1 image; out_u8 (W*H, optional) = normalised image;
 * *is_u8 = 1 when the reference would return CV_8UC1 (normalized && max > min). minmax optional [2]. */
int eorb_ev2im(eorb_ctx* ctx, const eorb_event* ev, size_t n, int W, int H, int pol, int normalized,
               float* out_f32, uint8_t* out_u8, float* minmax, int* is_u8);

/* replaces EvImConverter::ev2im_gauss, src/Event/EventConversion.cc:215-269
 * (include/Event/EventConversion.h:54-56; callers EvImBuilder.cpp:1345,1070) */
int eorb_ev2im_gauss(eorb_ctx* ctx, const eorb_event* ev, size_t n, int W, int H, float sigma, int pol,
                     int normalized, float* out_f32, uint8_t* out_u8, float* minmax);

/* ---- raw sensor events through the undistortion maps ------------------------------
 * mapX / mapY = MyCalibrator::mUndistMapX / mUndistMapY (Utils/MyCalibrator.cpp:60-101, LH x LW floats each, built by the
 * caller with cv::undistortPoints as the reference does).  checkInImage = the flag the loader passes to
 * getEventChunkRectified (EventLoader.cpp:264-305): events whose undistorted point fails MyCalibrator::isInImage(x, y)
 * for the accumulation image (:31-34) are dropped. */
int eorb_set_undistort_maps(eorb_ctx* ctx, const float* mapX, const float* mapY, int LW, int LH, int checkInImage);

/* replaces the rectification loop of EventDataStore::getEventChunkRectified (src/Event/EventLoader.cpp:264-305) after
 * parsing: out[k] = {raw.t / tsFactor, mapX[y][x], mapY[y][x], p} (MyCalibrator::undistPointMaps :164-180) for the events
 * kept by checkInImage against a W x H image, in order.  out has room for n events; *n_out = number kept. */
int eorb_undistort_events(eorb_ctx* ctx, const eorb_raw_event* raw, size_t n, int W, int H, double tsFactor,
                          eorb_event* out, size_t* n_out);

/* replaces the text half of the loader: getline + EventDataStore::parseLine ("stream >> ts >> x >> y >> p",
 * src/Event/EventLoader.cpp:80-92) + BaseLoader::isComment (Utils/DataStore.cpp:111-114) over a whole buffer of the dataset's
 * events.txt.  Accepted grammar per line: `ts x y p` as plain decimals (no exponent) separated by blanks / tabs, optional '\r';
 * ts with at most 19 significant digits, value < 2^53 / 10^frac and at most 22 fractional digits (then one IEEE division gives
 * strtod's result); x, y integer-valued in 0..65535 (the reference truncates them, MyCalibrator.cpp:172-173); p in {0, 1}.
 * '#' lines and blank lines are skipped.  Any other line: EORB_E_ARG with *bad_line = its 0-based index (the caller falls back
 * to its own parser for that file).  out has room for cap events. */
int eorb_parse_events_text(eorb_ctx* ctx, const char* text, size_t nbytes, eorb_raw_event* out, size_t cap, size_t* n_out,
                           int64_t* bad_line);

/* = eorb_undistort_events followed by eorb_ev2im_gauss / eorb_ev2im on the kept events, fused: the stamp of a sensor
 * pixel depends only on its map entry, so the (2h+1)^2 values per pixel are tabulated once per (maps, sigma) and the
 * accumulation kernel only orders and adds them.  Bit-identical to the two-step path. */
int eorb_ev2im_gauss_raw(eorb_ctx* ctx, const eorb_raw_event* raw, size_t n, int W, int H, float sigma, int pol,
                         int normalized, float* out_f32, uint8_t* out_u8, float* minmax);
int eorb_ev2im_raw(eorb_ctx* ctx, const eorb_raw_event* raw, size_t n, int W, int H, int pol, int normalized,
                   float* out_f32, uint8_t* out_u8, float* minmax, int* is_u8);

/* ---- motion-compensated accumulation (host buffers) ------------------------------------------------ */
typedef struct { float fx, fy, cx, cy; } eorb_pinhole;     /* Pinhole::mvParameters (CameraModels/Pinhole.cpp:30-62) */

/* replaces EvImConverter::ev2mci_gg_f(evs, pCamera, Tcw, medDepth, W, H, sigma, pol, normalized)
 * (src/Event/EventConversion.cc:280-360) and the depth-map overload (:451-531, pass depth_per_event[n] =
 * depthMapObj.getDepthLinInterp(ex, ey)).  angle / axis = Eigen::AngleAxisd(R of Tcw), t = translation of Tcw (the
 * adapter computes them once per call with Eigen, as the reference does at :297-301).  n == 0 -> zero image. */
int eorb_ev2mci_se3(eorb_ctx* ctx, const eorb_event* ev, size_t n, const eorb_pinhole* cam, double angle,
                    const double axis[3], const double t[3], float medDepth, const float* depth_per_event,
                    int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax);
/* replaces the SE2 overload (params2D = {omega, vx, vy[, scale]}), src/Event/EventConversion.cc:363-448 */
int eorb_ev2mci_se2(eorb_ctx* ctx, const eorb_event* ev, size_t n, const eorb_pinhole* cam, const float* params2D, int nparams,
                    int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax);
/* GeometricCamera of the motion-compensated images: model 0 = Pinhole (fx, fy, cx, cy), 1 = KannalaBrandt8 (+ k[0..3] =
 * mvParameters[4..7], precision = KB8_DEF_PRECISION 1e-6: include/CameraModels/KannalaBrandt8.h:36; the MVSEC configuration,
 * Examples/Event/EvMVSEC_ETHZ.yaml:54-67).  unproject / project follow src/CameraModels/KannalaBrandt8.cpp:87-190. */
typedef struct { int model; float fx, fy, cx, cy; float k[4]; float precision; } eorb_camera;
int eorb_ev2mci_se3_cam(eorb_ctx* ctx, const eorb_event* ev, size_t n, const eorb_camera* cam, double angle,
                        const double axis[3], const double t[3], float medDepth, const float* depth_per_event,
                        int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax);
int eorb_ev2mci_se2_cam(eorb_ctx* ctx, const eorb_event* ev, size_t n, const eorb_camera* cam, const float* params2D, int nparams,
                        int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax);
/* replaces EvImConverter::measureImageFocus (src/Event/EventConversion.cc:74-111) */
int eorb_measure_image_focus(eorb_ctx* ctx, const float* img, int W, int H, float* focus);
/* n images of W x H back to back in one call (the motion-compensation contest scores its reconstructions together,
 * src/Event/EvImBuilder.cpp:1165-1203): focus[n] */
int eorb_measure_image_focus_n(eorb_ctx* ctx, const float* imgs, int n, int W, int H, float* focus);
/* replaces cv::normalize(img, img, 255, 0, NORM_MINMAX, CV_8UC1) at src/Event/EvImBuilder.cpp:976,1055,1076,1140 */
int eorb_normalize_minmax_u8(eorb_ctx* ctx, const float* img, int W, int H, uint8_t* out);

/* ---- ORB extractor (host buffers) --------------------------------------------------------------- */
/* replaces ORBextractor::ORBextractor, src/ORBextractor.cc:420-489: scale tables, per-level quotas,
 * edge threshold (per context, not a process global: SURVEY App.B H3) for images of W x H */
int eorb_orb_configure(eorb_ctx* ctx, const eorb_orb_params* p, int W, int H);
int eorb_orb_max_keypoints(eorb_ctx* ctx);               /* output capacity needed */
int eorb_orb_get_tables(eorb_ctx* ctx, float* scale_factors, float* inv_scale_factors,
                        int* features_per_level, int* edge_threshold);   /* getters hdr:83-107 */

/* replaces ORBextractor::operator() with descriptors (src/ORBextractor.cc:1092-1176) and the
 * detect-only overload (:1178-1238); include/ORBextractor.h:75-81.  lap0/lap1 = vLappingArea.
 * kps[cap], desc[cap*32] (may be NULL when !want_desc), oob[cap] optional (1 = some rBRIEF tap of
 * that keypoint left the blurred level buffer: the reference reads out of bounds there, H4).
 * *mono_index = the reference's return value. */
int eorb_orb_extract(eorb_ctx* ctx, const uint8_t* img, int W, int H, int stride, int lap0, int lap1,
                     int want_desc, eorb_keypoint* kps, uint8_t* desc, uint8_t* oob, int cap,
                     int* n_out, int* mono_index);

/* replaces ORBextractor::ComputeTrackedKPtsDesc (src/ORBextractor.cc:1316-1363; callers EvAsynchTrackerU.cpp:794,812):
 * descriptor of every tracked keypoint at the pyramid level of its octave (pt * mvInvScaleFactor[octave], kp.angle).
 * desc: n x 32; rows whose octave is outside [0, nlevels) are zero (uninitialised in the reference). oob optional. */
int eorb_orb_tracked_descriptors(eorb_ctx* ctx, const uint8_t* img, int W, int H, int stride,
                                 const eorb_keypoint* kps, int n, uint8_t* desc, uint8_t* oob);

/* replaces ORBextractor::AssignKPtLevelByBestDesc (src/ORBextractor.cc:1267-1314; caller EvSynchTracker.cpp:790):
 * kps[i].octave <- the level whose descriptor at pt * invScale[level] is closest (Hamming) to ref_desc row i. */
int eorb_orb_assign_level_by_best_desc(eorb_ctx* ctx, const uint8_t* img, int W, int H, int stride,
                                       const uint8_t* ref_desc, eorb_keypoint* kps, int n);

/* ---- matchers (host buffers) ------------------------------------------------------------------------ */
/* replaces ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:714-831) and
 * MixedMatcher::SearchForInitialization (src/MixedMatcher.cpp:20-145; is_orb* = isORBDescValid gate,
 * NULL = all ORB).  kps*: undistorted keypoints; desc*: n x stride bytes, first 32 compared
 * (ORBmatcher.cc:2360-2378).  prev_matched: n1 (x,y) in/out; matches12: n1 out. */
int eorb_search_for_initialization(eorb_ctx* ctx,
        const eorb_keypoint* kps1, int n1, const uint8_t* desc1, int stride1, const uint8_t* is_orb1,
        const eorb_keypoint* kps2, int n2, const uint8_t* desc2, int stride2, const uint8_t* is_orb2,
        const eorb_grid_bounds* gb, float* prev_matched, int32_t* matches12,
        int windowSize, float nnratio, int checkOri, int* nmatches);

/* replaces the mono branch of ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)
 * (src/ORBmatcher.cc:1969-2187; MixedMatcher.cpp:693-).  Projection stays on the host:
 * valid/uv per last-frame keypoint, mp_desc (n_last x 32), mp_obs; cur_mp in/out
 * (-1 none, k>=0 last-frame point k, -2 foreign observed, -3 foreign unobserved).
 * mode 0: levels [o-1,o+1]; 1 forward (>= o); 2 backward ([0,o]). */
int eorb_search_by_projection_last(eorb_ctx* ctx,
        const eorb_keypoint* cur_kps, int n_cur, const uint8_t* cur_desc, int cur_stride, const uint8_t* cur_is_orb,
        const eorb_keypoint* last_kps, int n_last, const uint8_t* last_is_orb,
        const uint8_t* valid, const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
        const float* level_scale /* n_last: getORBScaleFactor(octave) or the AKAZE factor */,
        const eorb_grid_bounds* gb, int32_t* cur_mp, float th, int mode, int checkOri, int* nmatches);

/* replaces ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist)
 * (src/ORBmatcher.cc:2189-2312; MixedMatcher.cpp:928-1063; relocalisation, f3).  One query per pKF feature i:
 * valid[i] = map point present, !isBad(), not in sAlreadyFound, projection inside the image and distance gates passed
 * (:2207-2234, on the host); uv, pred_level = PredictScale, level_scale = getORBScaleFactor / getAKAZEScaleFactor of that level,
 * mp_desc = pMP->GetDescriptor().  cur_mp in/out: -1 free, anything else occupied; a match writes i (the pKF feature index). */
int eorb_search_by_projection_kf(eorb_ctx* ctx,
        const eorb_keypoint* cur_kps, int n_cur, const uint8_t* cur_desc, int cur_stride, const uint8_t* cur_is_orb,
        const eorb_keypoint* kf_kps, int n_kf, const uint8_t* kf_is_orb,
        const uint8_t* valid, const float* uv, const int32_t* pred_level, const float* level_scale, const uint8_t* mp_desc,
        const eorb_grid_bounds* gb, int32_t* cur_mp, float th, int ORBdist, int checkOri, int* nmatches);

/* replaces the mono branch of ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
 * (src/ORBmatcher.cc:44-219; MixedMatcher.cpp:500-691). */
int eorb_search_by_projection_map(eorb_ctx* ctx,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const uint8_t* is_orb,
        int M, const uint8_t* in_view, const float* proj_xy, const int32_t* level, const float* view_cos,
        const uint8_t* mp_desc, const uint8_t* mp_obs, const uint8_t* mp_is_orb, const float* level_scale,
        const eorb_grid_bounds* gb, int32_t* frame_mp, float th, float nnratio, int* nmatches);

/* The rectified-stereo / RGB-D gate of the two matchers above -- "if(F.mvuRight[idx]>0) { er = fabs(projXR - F.mvuRight[idx]);
 * if(er > radius) continue; }" (src/ORBmatcher.cc:96-104 with pMP->mTrackProjXR and r * getORBScaleFactor(level); :2056-2062 with
 * ur = uv.x - mbf * invzc and th * getORBScaleFactor(octave)): uright = mvuRight of the searched frame's keypoints (n floats, <= 0: no
 * right match), proj_xr / proj_ur = the right coordinate of every query, computed by the caller with the projection.  Everything else
 * as the mono entry points (whose configurations pass no mvuRight: Frame::numKPtsLeft() != -1 or all mvuRight = -1). */
int eorb_search_by_projection_map_stereo(eorb_ctx* ctx,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const uint8_t* is_orb,
        int M, const uint8_t* in_view, const float* proj_xy, const int32_t* level, const float* view_cos,
        const uint8_t* mp_desc, const uint8_t* mp_obs, const uint8_t* mp_is_orb, const float* level_scale,
        const eorb_grid_bounds* gb, int32_t* frame_mp, float th, float nnratio, const float* uright, const float* proj_xr, int* nmatches);
int eorb_search_by_projection_last_stereo(eorb_ctx* ctx,
        const eorb_keypoint* cur_kps, int n_cur, const uint8_t* cur_desc, int cur_stride, const uint8_t* cur_is_orb,
        const eorb_keypoint* last_kps, int n_last, const uint8_t* last_is_orb,
        const uint8_t* valid, const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
        const float* level_scale, const eorb_grid_bounds* gb, int32_t* cur_mp, float th, int mode, int checkOri,
        const float* cur_uright, const float* proj_ur, int* nmatches);

/* replaces the stereo constructor's hot path, Frame::Frame(imLeft, imRight, ...) (src/Frame.cc:97-152): ExtractORB on both images
 * (:122-125, two extractors of equal parameters, vLappingArea {0, 0}) and Frame::ComputeStereoMatches (:869-1048: row-band candidates,
 * best descriptor distance below (TH_HIGH + TH_LOW) / 2, 11 x 11 L1 correlation over the shifts -5..5 on the keypoint's level image,
 * parabola, median cut).  mb = baseline in metres (mbf / fx), mbf = baseline x fx.  Out: the keypoints and descriptors of both images
 * (mvKeys / mvKeysRight order), uRight / depth [nL] = mvuRight / mvDepth (-1: none), nmatches = correlated matches before the median
 * cut.  One upload, one wait, one download.  The rectified images are what the caller passes (the reference assumes undistorted
 * images here, :136-141). */
int eorb_frame_stereo(eorb_ctx* ctx, const uint8_t* imLeft, const uint8_t* imRight, int W, int H, int stride, float mb, float mbf,
                      eorb_keypoint* kpsL, uint8_t* descL, int* nL, eorb_keypoint* kpsR, uint8_t* descR, int* nR, int cap,
                      float* uRight, float* depth, int* nmatches);

/* replaces the mono branch of ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (src/ORBmatcher.cc:276-478;
 * MixedMatcher.cpp:148-356).  DBoW2::FeatureVector as CSR (node ids ascending, offsets, feature indices in vector
 * order).  kf_has_mp[i] = map point present and !isBad().  match_f[n_f] out = KeyFrame feature index or -1. */
int eorb_search_by_bow(eorb_ctx* ctx,
        const eorb_keypoint* kf_kps, int n_kf, const uint8_t* kf_desc, const uint8_t* kf_has_mp,
        const uint32_t* kf_nodes, const int32_t* kf_node_off, const int32_t* kf_idx, int kf_nn,
        const eorb_keypoint* f_kps, int n_f, const uint8_t* f_desc,
        const uint32_t* f_nodes, const int32_t* f_node_off, const int32_t* f_idx, int f_nn,
        int32_t* match_f, float nnratio, int checkOri, int* nmatches);

/* replaces the mono branch of ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (src/ORBmatcher.cc:833-973; loop
 * closing / place recognition).  match12[n1] out = index of the pKF2 feature, or -1. */
int eorb_search_by_bow_kf(eorb_ctx* ctx,
        const eorb_keypoint* kps1, int n1, const uint8_t* desc1, const uint8_t* has_mp1,
        const uint32_t* nodes1, const int32_t* node_off1, const int32_t* idx1, int nn1,
        const eorb_keypoint* kps2, int n2, const uint8_t* desc2, const uint8_t* has_mp2,
        const uint32_t* nodes2, const int32_t* node_off2, const int32_t* idx2, int nn2,
        int32_t* match12, float nnratio, int checkOri, int* nmatches);

/* replaces the mono branch of ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo=false, bCoarse)
 * (src/ORBmatcher.cc:975-1214; MixedMatcher.cpp:1326-1573; local mapping, f3).  elig1[i] = !pKF1->GetMapPoint(i) &&
 * isORBDescValid(i), elig2 likewise (vbMatched2 is never written by the reference).  ep[2] = the epipole
 * pKF2->mpCamera->project(R2w*Cw+t2w) (:982-987); F12[9] row-major = K1.t().inv()*t12x*R12*K2.inv(), the matrix
 * Pinhole::epipolarConstrain rebuilds for every candidate (Pinhole.cpp:137-140).  scale2[l] = pKF2->getORBScaleFactor(l),
 * sigma2_2[l] = pKF2->getORBLevelSigma2(l).  match12[n1] out = vMatches12 (the caller forms vMatchedPairs, :1203-1211).
 * Rectified stereo: bit 1 of elig1[i] / elig2[i] set = the keypoint has a right coordinate (bStereo1 / bStereo2: mvuRight >= 0, :1051,
 * :1079): the epipole-distance test (:1093-1100) is skipped for a pair with such a keypoint; bOnlyStereo = clear bit 0 of the others. */
int eorb_search_for_triangulation(eorb_ctx* ctx,
        const eorb_keypoint* kps1, int n1, const uint8_t* desc1, int stride1, const uint8_t* elig1,
        const uint32_t* nodes1, const int32_t* node_off1, const int32_t* idx1, int nn1,
        const eorb_keypoint* kps2, int n2, const uint8_t* desc2, int stride2, const uint8_t* elig2,
        const uint32_t* nodes2, const int32_t* node_off2, const int32_t* idx2, int nn2,
        const float* ep, const float* F12, const float* scale2, const float* sigma2_2, int nlevels,
        int bCoarse, int checkOri, int32_t* match12, int* nmatches);

/* replaces the search core shared by ORBmatcher::Fuse (src/ORBmatcher.cc:1512-1578 and :1700-1720), SearchBySim3
 * (:1829-1860, :1909-1940) and SearchByProjection(KeyFrame*, Scw, ...) (:548-588, :667-706): for every projected map point
 * m (valid[m], uv, radius = th*getORBScaleFactor(level), predicted level, descriptor) the best keypoint among
 * KeyFrame::GetFeaturesInArea(u, v, radius) (src/KeyFrame.cc:873-917) with octave in [level-1, level].  Projection and the
 * map update stay with the caller.
 *   inv_sigma2 != NULL : Fuse's mono reprojection gate e2*inv_sigma2[octave] > 5.99 (:1557-1564)
 *   taken != NULL      : n flags in/out, SearchByProjection(KF,Scw) semantics: queries in order, flagged keypoints skipped,
 *                        taken[best] = 1 when (float)best_dist <= accept_thr (= TH_LOW*ratioHamming, :582-586)
 * best_idx[m] = -1 and best_dist[m] = 256 when no keypoint qualifies. */
int eorb_kf_radius_match(eorb_ctx* ctx,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const eorb_grid_bounds* gb,
        int M, const uint8_t* valid, const float* uv, const float* radius, const int32_t* level, const uint8_t* q_desc,
        const float* inv_sigma2, int nlevels, uint8_t* taken, float accept_thr, int32_t* best_idx, int32_t* best_dist);
/* Fuse on a rectified-stereo KeyFrame (src/ORBmatcher.cc:1541-1553; the same in the Sim3 overload :1683-): a keypoint with a right
 * coordinate (uright[i] = pKF->mvuRight[i] >= 0) is gated by the three-term error ex^2 + ey^2 + (q_ur[m] - uright[i])^2 against 7.8
 * instead of the two-term one against 5.99; q_ur[m] = u - bf * invz of map point m.  inv_sigma2 must be given. */
int eorb_kf_radius_match_stereo(eorb_ctx* ctx,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const eorb_grid_bounds* gb,
        int M, const uint8_t* valid, const float* uv, const float* radius, const int32_t* level, const uint8_t* q_desc,
        const float* inv_sigma2, int nlevels, const float* uright, const float* q_ur, int32_t* best_idx, int32_t* best_dist);

/* replaces MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:349-423; f3), batched over M map points: the
 * descriptors observed for map point m are rows offsets[m] .. offsets[m+1]-1 of desc (n x 32); best[m] = the row (relative
 * to offsets[m]) with the least median Hamming distance to the others, -1 when there is none. */
int eorb_distinctive_descriptors(eorb_ctx* ctx, const uint8_t* desc, const int32_t* offsets, int M, int32_t* best);

/* ---- KLT tracker ----------------------------------------------------------------------------------
 * replaces cv::calcOpticalFlowPyrLK(mRefFrame, currImage, mRefPoints, kpts, status, err, Size(mPatchSz, mPatchSz), mMaxLevel,
 * mLKCriteria, flags) inside ELK_Tracker::trackCurrImage (src/Event/KLT_Tracker.cpp:49-98; Event.klt.* of
 * Examples/Event/EvETHZ.yaml:205-208: winSize 23, maxLevel 1, maxIter 10, eps 0.03).  8-bit single-channel images.
 * TermCriteria(COUNT + EPS, maxCount, epsilon); flags: 4 = OPTFLOW_USE_INITIAL_FLOW (next_pts holds the initial guess),
 * 8 = OPTFLOW_LK_GET_MIN_EIGENVALS; minEigThreshold = 1e-4 is OpenCV's default.  next_pts (n x 2) in/out, status / err (n) out.
 * The match bookkeeping around it (refineTrackedPts :104-151, refineFirstOctaveLevel :153-205) is host logic and stays in the
 * adapter. */
int eorb_calc_optical_flow_pyr_lk(eorb_ctx* ctx, const uint8_t* prev, const uint8_t* next, int W, int H, int stride,
                                  const float* prev_pts, float* next_pts, int n, int win, int maxLevel, int maxCount, double epsilon,
                                  int flags, float minEigThreshold, uint8_t* status, float* err);

/* ---- DBoW2 vocabulary transform: the producer of the feature vectors SearchByBoW consumes ----------
 * The vocabulary tree (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h: m_nodes after loadFromTextFile :1338-1430) flattened:
 * node 0 = root; children of node i = child_ids[child_off[i] .. child_off[i+1]) in `children` order; a node without children
 * is a word (Node::isLeaf) with word_id / weight; node_desc = nnodes x 32 bytes (FORB).  Copied to the device once. */
int eorb_bow_set_vocabulary(eorb_ctx* ctx, int nnodes, int L, const int32_t* child_off, const int32_t* child_ids,
                            const uint8_t* node_desc, const int32_t* word_id, const double* weight);

/* replaces ORBVocabulary::transform(vCurrentDesc, mBowVec, mFeatVec, levelsup) as called by Frame::ComputeBoW
 * (src/Frame.cc: levelsup = 4) = TemplatedVocabulary::transform :1125-1190 + the per-feature descent :1208-1250 with
 * FORB::distance (FORB.cpp:81-101).  weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (WeightingType); norm: 0 none, 1 L1, 2 L2 =
 * ScoringObject::mustNormalize of the vocabulary's scoring type (ORBvoc.txt: TF_IDF + L1_NORM -> 0, 1).
 * BowVector out: ascending (bow_word, bow_val)[*n_words]; FeatureVector out: CSR (fv_node ascending, fv_off[*n_fvnodes + 1],
 * fv_idx in push_back order) -- the layout eorb_search_by_bow takes.  Output arrays sized n (fv_off n + 1).
 * word_of / node_of (n, may be NULL): word and nid-level node of every feature, -1 for stopped words. */
int eorb_bow_transform(eorb_ctx* ctx, const uint8_t* desc, int n, int stride, int levelsup, int weighting, int norm,
                       uint32_t* bow_word, double* bow_val, int* n_words, uint32_t* fv_node, int32_t* fv_off, int32_t* fv_idx,
                       int* n_fvnodes, int32_t* word_of, int32_t* node_of);

/* The loop body every windowed matcher of src/ORBmatcher.cc shares (e.g. :754-774, :95-130, :2050-2070): for query q the
 * candidates cand_idx[cand_offsets[q] .. cand_offsets[q+1]) (what GetFeaturesInArea returned, after the caller's gates) are
 * visited in order with `if (d < best) {second = best; best = d} else if (d < second) second = d` on the first 32 descriptor
 * bytes.  For adapters that keep the greedy bookkeeping on the host.  Absent: index -1, distance 256. */
int eorb_hamming_window_match(eorb_ctx* ctx, const uint8_t* q_desc, int nq, int q_stride, const uint8_t* t_desc, int nt, int t_stride,
                              const int32_t* cand_offsets, const int32_t* cand_idx, int32_t* best_idx, int32_t* best_d,
                              int32_t* second_idx, int32_t* second_d);

/* replaces MixedFrame::sortFeaturesResponse (src/MixedFrame.cpp:211-225): perm[k] = index of the k-th keypoint in
 * descending-response order, equal responses in insertion order (std::multimap semantics). */
int eorb_sort_by_response(eorb_ctx* ctx, const eorb_keypoint* kps, int n, int32_t* perm);
/* MixedFrame::resolveNumMixedPts (src/MixedFrame.cpp:281-317): how many ORB / AKAZE features the mixed frame keeps */
void eorb_resolve_num_mixed(int nDetectedORB, int nDetectedAK, int nDesired, int nDesiredAK, int* nORB, int* nAK);

/* replaces cv::BFMatcher(NORM_HAMMING)::knnMatch(q, t, matches, 2) at src/Frame.cc:1228
 * (+ the ORBmatcher::DescriptorDistance core, ORBmatcher.cc:2360-2378).  idx2/dist2: nq*2. */
int eorb_hamming_bf_knn2(eorb_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                         int32_t* idx2, int32_t* dist2);

/* ---- batched, HBM-resident front end (throughput path) ------------------------------------------- */
typedef struct {
    int   W, H;
    float sigma;            /* Event.image.l1Sigma */
    int   pol;
    eorb_orb_params orb;
    int   lap0, lap1;       /* vLappingArea (mono: 0, 1000) */
    int   want_desc;
    int   max_batch;        /* slices per batch */
    int   max_events;       /* events per slice (capacity) */
    int   match;            /* 1: SearchForInitialization of slice b against slice b-1 */
    int   windowSize;       /* 100 */
    float nnratio;          /* 0.9 */
    int   checkOri;
} eorb_fe_config;

int eorb_fe_configure(eorb_ctx* ctx, const eorb_fe_config* cfg);

/* One pass of the hot path over a batch of B time-slices, everything resident in HBM:
 *   accumulate (ev2im_gauss, normalised u8) -> ORB extract -> match slice b against slice b-1
 *   (slice 0 against `prev` state carried in the ctx from the previous batch, if any).
 * d_events: all slices' events back to back; h_offsets[B+1]: slice boundaries (host array).
 * Outputs (device pointers, any may be NULL): d_images u8 B*W*H; d_kps B*cap; d_desc B*cap*32;
 * d_nkps int32 B; d_matches12 int32 B*cap (for slice b: index into slice b, per keypoint of b-1);
 * d_nmatches int32 B.  cap = eorb_orb_max_keypoints(). */
/* as eorb_fe_run_batch_dev, for raw sensor events resident in HBM (eorb_set_undistort_maps first) */
int eorb_fe_run_batch_raw_dev(eorb_ctx* ctx, const eorb_raw_event* d_events, const int64_t* h_offsets, int B,
                              uint8_t* d_images, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                              int32_t* d_matches12, int32_t* d_nmatches);

/* as eorb_fe_run_batch_raw_dev, for 4-byte sensor records (eorb_raw_event4) resident in HBM */
int eorb_fe_run_batch_raw4_dev(eorb_ctx* ctx, const eorb_raw_event4* d_events, const int64_t* h_offsets, int B,
                               uint8_t* d_images, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                               int32_t* d_matches12, int32_t* d_nmatches);

/* as eorb_fe_run_batch_raw_dev, for 2-byte sensor records (eorb_raw_event2) resident in HBM: polarity-free images on sensors of at
 * most 65 535 pixels (a DAVIS 240x180 has 43 200) never read more of an event than its pixel */
int eorb_fe_run_batch_raw2_dev(eorb_ctx* ctx, const eorb_raw_event2* d_events, const int64_t* h_offsets, int B,
                               uint8_t* d_images, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                               int32_t* d_matches12, int32_t* d_nmatches);

/* (float events: a call with 2^20 events or more waits for the stream once, to learn how many distinct positions its events take --
 * the raw variant above never waits) */
int eorb_fe_run_batch_dev(eorb_ctx* ctx, const eorb_event16* d_events, const int64_t* h_offsets, int B,
                          uint8_t* d_images, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                          int32_t* d_matches12, int32_t* d_nmatches);

/* ---- the L1 image builder's per-chunk path, one call per chunk ----------------------------------------------------------------
 * EvImBuilder::Track (src/Event/EvImBuilder.cpp:1300-1515) makes, per chunk of l1ChunkSize events, the event image (:1345) and a frame
 * from it (:1348); the frame of an INIT chunk runs the detect-only ORBextractor (EvBaseTracker::makeFrame -> EvFrame ctor,
 * src/Event/EventFrame.cpp:199-247) and becomes the reference of the LK tracker (init :568-592 -> ELK_Tracker::setRefImage), the frame
 * of a TRACKING chunk tracks the reference points into the new image (makeFrame :528-548 -> ELK_Tracker::trackAndMatchCurrImage,
 * src/Event/KLT_Tracker.cpp:215-234).  Through eorb_ev2im_gauss + eorb_orb_extract / eorb_calc_optical_flow_pyr_lk that is two or three
 * host-buffer calls, each with its own upload, wait and download, and the image crosses the link three times; the calls below take the
 * chunk's events and hand back keypoints / tracked points with ONE upload, ONE wait and ONE download.  The u8 image stays on the device
 * (out_u8 != NULL downloads it with the results; eorb_ev_slice_image fetches it later, until the next call on the context).
 * Events: `ev` (the reference's float EventData) or `raw` (sensor events resolved through eorb_set_undistort_maps), not both.
 * The image size is the extractor's (eorb_orb_configure). */
typedef struct eorb_klt_params {      /* Event.klt.* of Examples/Event/EvETHZ.yaml:205-208 -> ELK_Tracker (KLT_Tracker.cpp:14-20) */
    int    win;                       /* kltWinSize (23) */
    int    maxLevel;                  /* maxLevel (1) */
    int    maxCount;                  /* kltMaxItr (10) */
    double epsilon;                   /* kltEps (0.03) */
    float  minEigThreshold;           /* cv::calcOpticalFlowPyrLK's default 1e-4 */
} eorb_klt_params;

/* INIT chunk: ev2im_gauss(events, W, H, sigma) -> ORBextractor::operator() (both overloads: want_desc) -> the image and its keypoints
 * become the LK reference kept on the device.  Outputs as eorb_orb_extract. */
int eorb_ev_slice_extract(eorb_ctx* ctx, const eorb_event* ev, const eorb_raw_event* raw, size_t n, float sigma, int lap0, int lap1,
                          int want_desc, eorb_keypoint* kps, uint8_t* desc, uint8_t* oob, int cap, int* n_out, int* mono_index,
                          uint8_t* out_u8);

/* TRACKING chunk: ev2im_gauss(events) -> cv::calcOpticalFlowPyrLK(reference image, image, reference points, pts, status, err, win,
 * maxLevel, criteria, OPTFLOW_USE_INITIAL_FLOW) (ELK_Tracker::trackCurrImage, KLT_Tracker.cpp:49-74).  pts[2 * nref]: in = the last
 * tracked points (mLastTrackedPts: the reference points on the first tracking chunk), out = the tracked points; nref = the number of
 * keypoints the last eorb_ev_slice_extract returned.  The reference frame's pyramid and derivatives are built once per reference. */
int eorb_ev_slice_track(eorb_ctx* ctx, const eorb_event* ev, const eorb_raw_event* raw, size_t n, float sigma, const eorb_klt_params* klt,
                        float* pts, uint8_t* status, float* err, int nref, uint8_t* out_u8);

/* the u8 image of the context's last eorb_ev_slice_extract / _track call (W x H of the extractor), while no other host-buffer call
 * has run on the context since: the adapter's lazy download behind the cv::Mat seam */
int eorb_ev_slice_image(eorb_ctx* ctx, uint8_t* out_u8);

/* The reconstruction contest of a dispatch, EvImBuilder::generateMCImage (src/Event/EvImBuilder.cpp:1146-1247), in one call: for the
 * accumulated window `ev` (float EventData with their time stamps) the reconstructions
 *   0 "DP"  ev2mci_gg_f(evs, camera, Tcw, medDepth)     getDPoseMCI :958-979    present when dp != NULL
 *   1 "BA"  the same with the BA pose                   getBAMCI :1033-1058     present when ba != NULL
 *   2 "EH"  ev2im_gauss(evs, normalized = false)        getEvHist :1060-1079    always
 *   3 "Opt" ev2mci_gg_f(evs, camera, paramsSE2)         getAff2DMCI :1124-1143  present when se2_params != NULL
 * each with measureImageFocus and cv::normalize(img, img, 255, 0, NORM_MINMAX, CV_8UC1); the winner is the largest focus, the first
 * of equals in that order (MciInfo = std::multimap<float, PoseImagePtr, std::greater<float>>, include/Utils/Visualization.h:29); a
 * winning "EH" is replaced by the histogram of the later half of the window (:1214-1216).  The poses come from the optimisers (out of
 * this library's scope): AngleAxisd(R) of Tcw's rotation, its translation and the median depth, exactly what eorb_ev2mci_se3 takes.
 * focus[0..3] = the methods' focus (-1: absent), focus[4] = the later-half histogram's; *winner = the winning method; out_u8
 * (optional) = the winner's image; with `l2` (a context on the same device whose extractor is the L2 tracker's; the contexts belong to
 * the calling thread) the winner goes through its detect-only extraction (isMcImageGood :260-267 = the L2 frame): kps / cap / n_out.
 * No events: returns with *winner = -1 (:1149-1152). */
typedef struct eorb_se3_motion { double angle; double axis[3]; double t[3]; float medDepth; } eorb_se3_motion;
int eorb_ev_mc_contest(eorb_ctx* ctx, const eorb_event* ev, size_t n, const eorb_camera* cam, const eorb_se3_motion* dp,
                       const eorb_se3_motion* ba, const float* se2_params, int nparams, int W, int H, float sigma, float focus[5],
                       int* winner, uint8_t* out_u8, eorb_ctx* l2, int lap0, int lap1, eorb_keypoint* kps, int cap, int* n_out);

/* The float images of the context's last eorb_fe_run_batch_*_dev call (what ev2im_gauss(..., normalized = false) returns,
 * src/Event/EventConversion.cc:264-268: the CV_32FC1 sums before normaliseImage): *d_f32 = device pointer to B x H x W floats, valid
 * until the next batch call on the context; h_minmax (optional, 2 * B floats: min, max per slice = the running extremes of
 * resolveMinMaxVals :32-39) is downloaded, which waits for the stream.  For callers that want the un-normalised image of a slice
 * (EvImBuilder::getEvHist, src/Event/EvImBuilder.cpp:1060-1079) and for bit-level verification of a batch. */
int eorb_fe_last_f32_dev(eorb_ctx* ctx, const float** d_f32, float* h_minmax, int B);

/* the same pass for B camera frames resident in HBM (u8, W x H each, back to back; the image side of the front end,
 * Frame::ExtractORB src/Frame.cc:467-482 + SearchForInitialization of frame b against frame b-1): no accumulation, the frames
 * are only read.  eorb_fe_configure as above (sigma / pol / max_events unused). */
int eorb_fe_run_batch_images_dev(eorb_ctx* ctx, const uint8_t* d_images, int B, eorb_keypoint* d_kps, uint8_t* d_desc,
                                 int32_t* d_nkps, int32_t* d_matches12, int32_t* d_nmatches);

/* self-check used by tests: number of floats v in [lo, hi] (0 < lo <= hi) for which the reciprocal+fma quotient the
 * accumulation kernel uses for v / (2*pi*sigma^2) differs from the IEEE-754 quotient (must be 0). */
int eorb_selfcheck_division(eorb_ctx* ctx, float lo, float hi, float sigma, uint64_t* mismatches);

/* self-check used by tests: order-independent 64-bit hash of a device math function over every float whose bit pattern
 * lies in [lo_bits, hi_bits]: which = 0 exp(-x) as used by exp_XY2f, 1 sin(x), 2 cos(x) as used by computeOrbDescriptor.
 * The CPU oracle computes the same hash of its own functions: equality proves the two agree on every input. */
int eorb_selfcheck_math(eorb_ctx* ctx, int which, uint32_t lo_bits, uint32_t hi_bits, uint64_t* hash);

/* helpers: convert host AoS events to the compact record; device alloc/copy without a HIP binding */
void  eorb_pack_events(const eorb_event* ev, size_t n, eorb_event16* out);
void* eorb_dev_alloc(eorb_ctx* ctx, size_t bytes);
int   eorb_dev_free(eorb_ctx* ctx, void* p);
int   eorb_dev_upload(eorb_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int   eorb_dev_download(eorb_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
