/*
 * eorb_oracle.h -- CPU ORACLE for the EORB-SLAM event-frame front end.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a strict-IEEE, single-threaded C restatement of the
 * reference's CPU algorithm for the hot path (event->image accumulation, ORB extraction,
 * 256-bit Hamming matching).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (eorb_slam_amd/, include/) never includes,
 * links or calls anything in oracle/.
 *
 * PARITY PINNING STATUS (see oracle/README.md):
 *   - the reference ships no tests / golden vectors / fixtures for this path, and its sources
 *     cannot be compiled here (they need OpenCV 3.4.1, Eigen, glog, boost ... none present).
 *   - reference-OWNED arithmetic (event splat, octree, IC_Angle moments, rBRIEF taps,
 *     DescriptorDistance, matcher control flow, grid) is restated line by line with file:line
 *     citations and checked against hand-derived known answers from the reference source.
 *   - OpenCV 3.4.1 arithmetic (resize, GaussianBlur, FAST, fastAtan2, convertTo, copyMakeBorder)
 *     and libm (expf, sinf, cosf) are restated from their published algorithms:
 *     "PARITY UNPINNED" at that boundary (no reference-side vectors exist to pin them).
 *     expf/sinf/cosf are additionally checked exhaustively against the host glibc.
 *
 * All citations are file:line relative to the reference repository root.
 */
#ifndef EORB_ORACLE_H
#define EORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data types ------------------------------------------------------------------------- */

/* include/Event/EventData.h:36-58 : {double ts; float x; float y; bool p} = 24 B with padding */
typedef struct {
    double  ts;
    float   x, y;
    uint8_t p;
    uint8_t pad_[7];
} orc_event;

/* cv::KeyPoint layout (28 B): pt.x pt.y size angle response octave class_id */
typedef struct {
    float   x, y;
    float   size;
    float   angle;
    float   response;
    int32_t octave;
    int32_t class_id;
} orc_keypoint;

/* include/ORBextractor.h:33-47 ORBxParams */
typedef struct {
    int   nfeatures;
    float scaleFactor;
    int   nlevels;
    int   iniThFAST;
    int   minThFAST;
    int   edgeTh;      /* <0 : adaptive 19*(imWidth/752) rule, src/ORBextractor.cc:481-488 */
    int   imWidth;     /* only used when edgeTh < 0 */
} orc_orb_params;

/* ---- math primitives (exposed for tests) ----------------------------------------------- */
float orc_expf(float x);            /* glibc>=2.28 expf algorithm, strict IEEE double, no FMA   */
float orc_sinf(float x);            /* glibc>=2.28 sinf algorithm (valid for |x| < 120)         */
float orc_cosf(float x);
float orc_tanf(float x);            /* glibc>=2.28 tanf (valid for |x| < 120)                   */
float orc_atanf(float x);           /* glibc atanf / atan2f (fdlibm float)                      */
float orc_atan2f(float y, float x);
float orc_fast_atan2(float y, float x);   /* OpenCV 3.4 cv::fastAtan2 polynomial, degrees     */
int   orc_cvround(double v);              /* cvRound: round-half-to-even                        */
/* order-independent 64-bit hash of f over every float whose bit pattern is in [lo_bits, hi_bits] (which: 0 expf(-x),
 * 1 sinf(x), 2 cosf(x), 3 tanf(x), 4 atanf(x)); the device computes the same hash of its own functions (eorb_selfcheck_math). */
uint64_t orc_math_hash(int which, uint32_t lo_bits, uint32_t hi_bits);
/* atan2f over the generated pairs first .. first + count - 1 (orc_atan2_pair: every other pair raw bit patterns, the others small
 * rationals); which = 5 of eorb_selfcheck_math */
void orc_atan2_pair(uint64_t i, float* y, float* x);
uint64_t orc_atan2_hash(uint64_t first, uint64_t count);

/* ---- event accumulation: src/Event/EventConversion.cc ------------------------------------ */

/* EvImConverter::ev2im (:173-212).  out_f32: W*H floats (accumulated image BEFORE normalisation,
 * always written).  out_u8: W*H bytes, written only when the reference would normalise.
 * minmax[0]=minVal, minmax[1]=maxVal (running values).  Returns 1 if the reference returns a
 * CV_8UC1 image (normalized && max>min), else 0 (CV_32FC1). */
int orc_ev2im(const orc_event* ev, size_t n, int W, int H, int pol, int normalized,
              float* out_f32, uint8_t* out_u8, float* minmax);

/* EvImConverter::ev2im_gauss (:215-269).  Same outputs; returns `normalized`. */
int orc_ev2im_gauss(const orc_event* ev, size_t n, int W, int H, float sigma, int pol,
                    int normalized, float* out_f32, uint8_t* out_u8, float* minmax);

/* ---- motion-compensated accumulation: src/Event/EventConversion.cc:280-531 -------------------- */
/* double-precision sin/cos used by the per-event AngleAxisd -> rotation matrix (Eigen): fdlibm kernels + two-term pi/2
 * reduction, strict IEEE double, valid for |x| < 100 (parity unpinned against glibc's double sin/cos; agrees to 1 ulp) */
double orc_dsin(double x);
double orc_dcos(double x);

/* raw sensor event (dataset line "ts x y p", EventLoader.cpp:80-92) */
typedef struct { uint16_t x, y; uint32_t p; double t; } orc_raw_event;

/* EventDataStore::getEventChunkRectified (src/Event/EventLoader.cpp:264-305) after parsing, for n raw events:
 * MyCalibrator::undistPointMaps (Utils/MyCalibrator.cpp:164-180: dst = (mapX[y][x], mapY[y][x]) with x = (int)src.x),
 * ts / tsFactor (:120-123), checkInImage -> MyCalibrator::isInImage against W x H (:31-34).  Returns the number kept. */
size_t orc_undistort_events(const orc_raw_event* raw, size_t n, const float* mapX, const float* mapY, int LW, int LH,
                            int W, int H, int checkInImage, double tsFactor, orc_event* out);

/* Text half of EventDataStore::getEventChunkRectified / getTxtData (src/Event/EventLoader.cpp:80-92, :264-305): one event per
 * line "ts x y p"; lines whose first non-blank character is '#' are comments (BaseLoader::isComment, Utils/DataStore.cpp:111-114).
 * ts: strtod (istringstream >> double); x, y: strtof then static_cast<int> (MyCalibrator.cpp:172-173); p: 0 / 1.
 * Accepted grammar (anything else -> returns -(line number) - 1, 0-based): decimal numbers without exponent, separated by
 * blanks / tabs, optional trailing '\r'; x and y integer-valued, 0..65535.  Blank lines are skipped.  Returns the number of events. */
long orc_parse_events_text(const char* text, size_t nbytes, orc_raw_event* out, size_t cap);

typedef struct { float fx, fy, cx, cy; } orc_pinhole;          /* Pinhole::mvParameters (float), CameraModels/Pinhole.cpp */
/* GeometricCamera: model 0 = Pinhole, 1 = KannalaBrandt8 (mvParameters[4..7] = k1..k4, precision = KB8_DEF_PRECISION 1e-6) */
typedef struct { int model; float fx, fy, cx, cy; float k[4]; float precision; } orc_camera;
void orc_mci_warp_se3_cam(const orc_event* ev, size_t n, const orc_camera* cam, double angle, const double axis[3],
                          const double tt[3], float medDepth, const float* depth_per_event, float* uv_out);
void orc_mci_warp_se2_cam(const orc_event* ev, size_t n, const orc_camera* cam, const float* params2D, int nparams, float* uv_out);
int orc_ev2mci_se3_cam(const orc_event* ev, size_t n, const orc_camera* cam, double angle, const double axis[3], const double tt[3],
                       float medDepth, const float* depth_per_event, int W, int H, float sigma, int pol, int normalized,
                       float* out_f32, uint8_t* out_u8, float* minmax);
int orc_ev2mci_se2_cam(const orc_event* ev, size_t n, const orc_camera* cam, const float* params2D, int nparams,
                       int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax);
/* sinf / cosf for any finite angle of interest here (|x| < 120; negative arguments by symmetry) */
float orc_sinf_any(float x);
float orc_cosf_any(float x);

/* per-event warp of ev2mci_gg_f(evs, cam, Tcw, medDepth, ...) (:304-335): angle/axis = Eigen::AngleAxisd(R) and tt = t of
 * Tcw, computed by the caller (host, once per call).  depth_per_event != NULL replaces medDepth per event (depth-map
 * variant :451-531).  uv_out: n x 2 floats (the arguments of breakFloatCoords). */
void orc_mci_warp_se3(const orc_event* ev, size_t n, const orc_pinhole* cam, double angle, const double axis[3],
                      const double tt[3], float medDepth, const float* depth_per_event, float* uv_out);
/* per-event warp of the SE2 variant (:363-412): params2D = {omega, vx, vy[, scale]} (nparams 3 or 4) */
void orc_mci_warp_se2(const orc_event* ev, size_t n, const orc_pinhole* cam, const float* params2D, int nparams, float* uv_out);
/* full ev2mci_gg_f: warp + the ev2im_gauss splat on the warped coordinates; n == 0 -> zeros, returns 0 (:292-295) */
int orc_ev2mci_se3(const orc_event* ev, size_t n, const orc_pinhole* cam, double angle, const double axis[3], const double tt[3],
                   float medDepth, const float* depth_per_event, int W, int H, float sigma, int pol, int normalized,
                   float* out_f32, uint8_t* out_u8, float* minmax);
int orc_ev2mci_se2(const orc_event* ev, size_t n, const orc_pinhole* cam, const float* params2D, int nparams,
                   int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax);
/* EvImConverter::measureImageFocus (:74-111): mean over 30x30 patches of the patch standard deviation (cv::meanStdDev,
 * double accumulation in raster order) */
float orc_measure_image_focus(const float* img, int W, int H);
/* cv::normalize(img, img, 255, 0, NORM_MINMAX, CV_8UC1) (EvImBuilder.cpp:1076) */
void orc_cv_normalize_minmax_u8(const float* img, size_t npix, uint8_t* dst);

/* normalizeImage (:67-72) == Mat::convertTo(CV_8UC1, alpha, beta) */
void orc_normalize_u8(const float* src, size_t npix, float maxVal, float minVal, uint8_t* dst);

/* ---- ORB extractor: src/ORBextractor.cc --------------------------------------------------- */
typedef struct orc_orb orc_orb;

orc_orb* orc_orb_create(const orc_orb_params* p);
void     orc_orb_destroy(orc_orb* e);

/* ctor tables (:420-489) */
int          orc_orb_edge_threshold(const orc_orb* e);
const float* orc_orb_scale_factors(const orc_orb* e);      /* nlevels */
const float* orc_orb_inv_scale_factors(const orc_orb* e);
const int*   orc_orb_features_per_level(const orc_orb* e);
const int*   orc_orb_umax(const orc_orb* e);               /* 16 entries */
int          orc_orb_max_keypoints(const orc_orb* e);      /* safe output capacity */

/* ORBextractor::operator() (:1092-1176 with descriptors, :1178-1238 detect only).
 * kps: capacity cap; desc: cap*32 (may be NULL if !want_desc); oob: cap bytes or NULL, set to 1
 * for keypoints for which at least one of the 512 rBRIEF taps falls outside the blurred level
 * buffer (the reference reads out of its allocation there: undefined; oracle substitutes 0).
 * Returns monoIndex (>=0) or -1 for an empty image (:1096), -2 bad config (H14), -3 cap too small. */
int orc_orb_extract(orc_orb* e, const uint8_t* img, int W, int H, int stride, int lap0, int lap1,
                    int want_desc, orc_keypoint* kps, uint8_t* desc, uint8_t* oob, int cap,
                    int* n_out);

/* stage introspection after the last orc_orb_extract() call (for stage-by-stage GPU parity) */
int            orc_orb_level_size(const orc_orb* e, int level, int* w, int* h);   /* ROI size */
const uint8_t* orc_orb_level_buffer(const orc_orb* e, int level, int* bw, int* bh);/* bordered */
const uint8_t* orc_orb_level_blur(const orc_orb* e, int level);                    /* w*h, may be NULL */
int            orc_orb_level_candidates(const orc_orb* e, int level, const orc_keypoint** out);
int            orc_orb_level_keypoints(const orc_orb* e, int level, const orc_keypoint** out);

/* ORBextractor::ComputeTrackedKPtsDesc (:1316-1363): descriptor of each tracked keypoint at the pyramid level given by
 * its octave (pt * mvInvScaleFactor[octave], kp.angle).  desc: n x 32 (rows of keypoints whose octave is outside
 * [0, nlevels) are left zero; the reference leaves them uninitialised).  oob: n flags or NULL. */
int orc_orb_tracked_descriptors(orc_orb* e, const uint8_t* img, int W, int H, int stride, const orc_keypoint* kps, int n,
                                uint8_t* desc, uint8_t* oob);
/* ORBextractor::AssignKPtLevelByBestDesc (:1267-1314): kps[i].octave = level with the smallest Hamming distance between
 * ref_desc row i and the descriptor computed at that level (first minimum wins). */
int orc_orb_assign_level_by_best_desc(orc_orb* e, const uint8_t* img, int W, int H, int stride, const uint8_t* ref_desc,
                                      orc_keypoint* kps, int n);

/* pieces exposed for unit tests */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride);
void orc_gaussian_blur5_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
void orc_gauss_kernel_q8(int ksize, double sigma, int* out);
/* cv::FAST(img, kps, threshold, true) TYPE_9_16; returns count, kps as (x,y,score) int triples */
int  orc_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, int* xys, int cap);
/* DistributeOctTree (:558-782); in: candidates, out: selected (capacity cap) */
int  orc_distribute_octree(const orc_keypoint* in, int n, int minX, int maxX, int minY, int maxY,
                           int N, orc_keypoint* out, int cap);
float orc_ic_angle(const uint8_t* center, int step, const int* umax);
int  orc_orb_descriptor(const uint8_t* img, int w, int h, int step, float kx, float ky, float angle_deg,
                        uint8_t* desc32);   /* returns 1 if any tap was out of the buffer */

/* ---- matchers: src/ORBmatcher.cc, src/MixedMatcher.cpp, src/Frame.cc ------------------------- */

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b);   /* ORBmatcher.cc:2360-2378 */
void orc_three_maxima(const int* sizes, int L, int* ind1, int* ind2, int* ind3); /* :2314-2355 */

/* Frame grid (src/Frame.cc:431-460, 710-793; include/Frame.h:45-46) */
typedef struct {
    float minX, minY, maxX, maxY;     /* mnMinX.. */
    float invW, invH;                 /* mfGridElementWidthInv/HeightInv */
} orc_grid_bounds;

typedef struct orc_frame orc_frame;
/* kps: undistorted keypoints (pt, octave, angle used); desc: N x desc_stride bytes (first 32 used);
 * is_orb: N flags or NULL (all ORB) -- MixedFrame type gate. */
orc_frame* orc_frame_create(const orc_keypoint* kps, int N, const uint8_t* desc, int desc_stride,
                            const uint8_t* is_orb, const orc_grid_bounds* gb);
void       orc_frame_destroy(orc_frame* f);
void       orc_grid_bounds_for_image(int W, int H, orc_grid_bounds* gb);  /* Frame.cc:862-866,362-363 */
int        orc_get_features_in_area(const orc_frame* f, float x, float y, float r,
                                    int minLevel, int maxLevel, int* out, int cap);

/* ORBmatcher::SearchForInitialization (:714-831) / MixedMatcher (:20-145).
 * prev_matched: N1 (x,y) pairs in/out; matches12: N1 out. returns nmatches */
int orc_search_for_initialization(const orc_frame* F1, const orc_frame* F2, float* prev_matched,
                                  int* matches12, int windowSize, float nnratio, int checkOri);

/* ORBmatcher::SearchByProjection(Frame& cur, const Frame& last, th, bMono) (:1969-2187), mono
 * branch.  Host-side projection results are inputs: for each last-frame keypoint i:
 *   valid[i]  : has a map point, not outlier, invzc>=0 and uv inside the image bounds
 *   uv[2i..]  : projected position;  octave/angle come from `last`;  mp_desc: N_last x 32
 *   mp_obs[i] : 1 if that map point has Observations()>0
 * cur_mp (N_cur, in/out): -1 = no map point; k>=0 = holds last-frame point k; -2 = holds a foreign
 * map point with Observations()>0; -3 = foreign map point without observations.
 * mode: 0 = [oct-1, oct+1], 1 = forward (>= oct), 2 = backward ([0, oct]).  returns nmatches */
int orc_search_by_projection_last(const orc_frame* cur, const orc_frame* last, const uint8_t* valid,
                                  const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
                                  int* cur_mp, float th, int mode, int checkOri,
                                  const float* level_scale /* per last kp: getORBScaleFactor(octave) */);

/* ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) (:44-219) mono branch.
 * Per map point m: in_view, projX, projY, level (mnTrackScaleLevel), viewCos, desc, obs flag,
 * level_scale[m] = F.getORBScaleFactor(level) (or the AKAZE factor, MixedMatcher.cpp:529-535).
 * frame_mp (N, in/out) as above with k = map point index. returns nmatches */
int orc_search_by_projection_map(const orc_frame* F, int M, const uint8_t* in_view, const float* proj_xy,
                                 const int* level, const float* view_cos, const uint8_t* mp_desc,
                                 const uint8_t* mp_obs, const uint8_t* mp_is_orb, int* frame_mp,
                                 float th, float nnratio, const float* level_scale);
/* the rectified-stereo gate of the two projection matchers (src/ORBmatcher.cc:96-104, :2056-2062) */
int orc_search_by_projection_last_stereo(const orc_frame* cur, const orc_frame* last, const uint8_t* valid,
                                         const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
                                         int* cur_mp, float th, int mode, int checkOri, const float* level_scale,
                                         const float* uright, const float* proj_ur);
int orc_search_by_projection_map_stereo(const orc_frame* F, int M, const uint8_t* in_view, const float* proj_xy,
                                        const int* level, const float* view_cos, const uint8_t* mp_desc,
                                        const uint8_t* mp_obs, const uint8_t* mp_is_orb, int* frame_mp,
                                        float th, float nnratio, const float* level_scale, const float* uright, const float* proj_xr);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (:276-478), mono branch.  The DBoW2 feature vectors
 * are CSR: node ids ascending, node_off[nn+1], idx[] = feature indices in vector order.  kf_has_mp[i] = map point present
 * and not bad.  match_f[N_F] out: KeyFrame feature index matched to each frame feature, or -1.  returns nmatches */
int orc_search_by_bow(const orc_keypoint* kf_kps, int n_kf, const uint8_t* kf_desc, const uint8_t* kf_has_mp,
                      const uint32_t* kf_nodes, const int32_t* kf_node_off, const int32_t* kf_idx, int kf_nn,
                      const orc_keypoint* f_kps, int n_f, const uint8_t* f_desc,
                      const uint32_t* f_nodes, const int32_t* f_node_off, const int32_t* f_idx, int f_nn,
                      int32_t* match_f, float nnratio, int checkOri);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) (:833-973), mono.  has_mp1 / has_mp2: map point present
 * and not bad.  match12[n1] out: index of the pKF2 feature whose map point is assigned to feature idx1, or -1. */
int orc_search_by_bow_kf(const orc_keypoint* kps1, int n1, const uint8_t* desc1, const uint8_t* has_mp1,
                         const uint32_t* nodes1, const int32_t* off1, const int32_t* idx1, int nn1,
                         const orc_keypoint* kps2, int n2, const uint8_t* desc2, const uint8_t* has_mp2,
                         const uint32_t* nodes2, const int32_t* off2, const int32_t* idx2, int nn2,
                         int32_t* match12, float nnratio, int checkOri);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (:2189-2312;
 * MixedMatcher.cpp:928-1063) after projection: query i = pKF feature i (valid, uv, nPredictedLevel, th-free level scale,
 * map-point descriptor, KeyFrame keypoint angle / type).  cur_mp: -1 free, anything else occupied; matches store i. */
int orc_search_by_projection_kf(const orc_frame* cur, const orc_keypoint* kf_kps, int n_kf, const uint8_t* kf_is_orb,
                                const uint8_t* valid, const float* uv, const int32_t* pred_level, const float* level_scale,
                                const uint8_t* mp_desc, int* cur_mp, float th, int ORBdist, int checkOri);

/* mono branch of ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo=false, bCoarse) (:975-1214)
 * with the MixedMatcher gate (MixedMatcher.cpp:1326-1573) folded into elig: elig1[i] = !GetMapPoint(i) && isORBDescValid(i).
 * ep = pKF2->mpCamera->project(R2w*Cw+t2w) and F12 = K1.t().inv()*t12x*R12*K2.inv() (Pinhole.cpp:137-140, constant over the
 * call) are computed by the caller.  scale2[l] = pKF2->getORBScaleFactor(l), sigma2_2[l] = pKF2->getORBLevelSigma2(l).
 * match12[n1] out (vMatches12). */
int orc_search_for_triangulation(const orc_keypoint* kps1, int n1, const uint8_t* desc1, int stride1, const uint8_t* elig1,
                                 const uint32_t* nodes1, const int32_t* off1, const int32_t* idx1, int nn1,
                                 const orc_keypoint* kps2, int n2, const uint8_t* desc2, int stride2, const uint8_t* elig2,
                                 const uint32_t* nodes2, const int32_t* off2, const int32_t* idx2, int nn2,
                                 const float ep[2], const float F12[9], const float* scale2, const float* sigma2_2,
                                 int bCoarse, int checkOri, int32_t* match12);

/* the search core shared by ORBmatcher::Fuse (:1512-1578, :1619-1741), SearchBySim3 (:1829-1860, :1909-1940) and
 * SearchByProjection(KeyFrame*, Scw, ...) (:548-588): for query m (a projected map point: uv, radius = th*scaleFactor(level),
 * predicted level, descriptor) the best keypoint of the KeyFrame among KeyFrame::GetFeaturesInArea(u,v,radius)
 * (KeyFrame.cc:873-917) with octave in [level-1, level].
 *  inv_sigma2 != NULL : Fuse's mono reprojection gate (:1557-1564) e2*inv_sigma2[octave] > 5.99 -> skip
 *  taken != NULL      : SearchByProjection(KF,Scw) semantics: queries in order, candidates with taken[idx] skipped, and
 *                       taken[bestIdx] = 1 when (float)bestDist <= accept_thr (:582-586)
 * best_idx[m] = -1 / best_dist[m] = 256 when nothing qualifies (the callers' INT_MAX start behaves the same). */
void orc_kf_radius_match(const orc_frame* kf, int M, const uint8_t* valid, const float* uv, const float* radius,
                         const int32_t* level, const uint8_t* q_desc, const float* inv_sigma2, uint8_t* taken,
                         float accept_thr, int32_t* best_idx, int32_t* best_dist);
void orc_kf_radius_match_stereo(const orc_frame* kf, int M, const uint8_t* valid, const float* uv, const float* radius,
                         const int32_t* level, const uint8_t* q_desc, const float* inv_sigma2,
                         const float* uright, const float* q_ur, int32_t* best_idx, int32_t* best_dist);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:349-423) for M map points: descriptors of map point m are rows
 * offsets[m]..offsets[m+1]-1 of desc (n x 32); best[m] = row (relative to offsets[m]) with the least median distance to
 * the others (first minimum), or -1 when the map point has no descriptor. */
void orc_distinctive_descriptors(const uint8_t* desc, const int32_t* offsets, int M, int32_t* best);

/* cv::calcOpticalFlowPyrLK as ELK_Tracker::trackCurrImage calls it (src/Event/KLT_Tracker.cpp:49-98): see orc_klt.c.
 * flags: 4 = OPTFLOW_USE_INITIAL_FLOW, 8 = OPTFLOW_LK_GET_MIN_EIGENVALS.  next_pts (n x 2) in/out, status / err (n) out. */
void orc_calc_optical_flow_pyr_lk(const uint8_t* prev, const uint8_t* next, int W, int H, int stride, const float* prev_pts,
                                  float* next_pts, int n, int win, int maxLevel, int maxCount, double epsilon, int flags,
                                  float minEigThreshold, uint8_t* status, float* err);

/* DBoW2 vocabulary tree (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h: m_nodes) flattened: node 0 = root; children of node i =
 * child_ids[child_off[i] .. child_off[i+1]) in `children` order; a node without children is a word (isLeaf) with word_id / weight. */
typedef struct {
    int nnodes, L;
    const int32_t* child_off;      /* nnodes + 1 */
    const int32_t* child_ids;
    const uint8_t* node_desc;      /* nnodes x 32 (FORB) */
    const int32_t* word_id;        /* nnodes */
    const double*  weight;         /* nnodes */
} orc_vocabulary;

/* TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (:1125-1190) with the per-feature descent
 * (:1208-1250; FORB::distance FORB.cpp:81-101).  weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; norm: 0 none, 1 L1, 2 L2
 * (ScoringObject::mustNormalize).  BowVector out as ascending (bow_word, bow_val)[*n_words]; FeatureVector out as CSR
 * (fv_node ascending, fv_off, fv_idx in push_back order)[*n_fvnodes].  Arrays sized n (fv_off n + 1).
 * word_of / node_of (n, optional): the word and the nid-level node every feature fell into (-1 for stopped words). */
void orc_bow_transform(const orc_vocabulary* voc, const uint8_t* desc, int n, int stride, int levelsup, int weighting, int norm,
                       uint32_t* bow_word, double* bow_val, int* n_words, uint32_t* fv_node, int32_t* fv_off, int32_t* fv_idx,
                       int* n_fvnodes, int32_t* word_of, int32_t* node_of);

/* the inner loop every windowed matcher of ORBmatcher.cc shares (e.g. :754-774): per query the candidates
 * cand_idx[cand_offsets[q] .. cand_offsets[q+1]) are visited in order with `if (d < best) {second = best; best = d; idx = c}
 * else if (d < second) second = d`.  best_idx / second_idx = -1 and distances = 256 when absent.  First 32 bytes of each row. */
void orc_hamming_window_match(const uint8_t* q_desc, int nq, int q_stride, const uint8_t* t_desc, int t_stride,
                              const int32_t* cand_offsets, const int32_t* cand_idx, int32_t* best_idx, int32_t* best_d,
                              int32_t* second_idx, int32_t* second_d);

/* MixedFrame::sortFeaturesResponse (MixedFrame.cpp:211-225): order = descending response, equal responses keep their
 * insertion order (multimap).  perm[k] = source index of the k-th output element. */
void orc_sort_by_response(const orc_keypoint* kps, int n, int32_t* perm);
/* MixedFrame::resolveNumMixedPts (MixedFrame.cpp:281-317) */
void orc_resolve_num_mixed(int nDetectedORB, int nDetectedAK, int nDesired, int nDesiredAK, int* nORB, int* nAK);

/* cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, 2) (Frame.cc:1228): per query two best (idx, dist)
 * ascending; ties -> lowest train index.  idx2/dist2: nq*2 (-1 / INT_MAX when nt < k). */
void orc_bf_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2, int32_t* dist2);

/* Frame::ComputeStereoMatches (src/Frame.cc:869-1048): eL / eR = the extractors that produced the left / right keypoints (their level
 * images are read); uRight / depth [N] out (-1: no match); returns the number of correlated matches before the median cut. */
int orc_compute_stereo_matches(const orc_orb* eL, const orc_orb* eR, const orc_keypoint* kL, int N, const uint8_t* dL,
                               const orc_keypoint* kR, int Nr, const uint8_t* dR, float mb, float mbf, float* uRight, float* depth);

#ifdef __cplusplus
}
#endif
#endif
