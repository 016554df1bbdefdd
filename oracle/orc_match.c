/*
 * orc_match.c -- ORACLE (test infrastructure only): 256-bit Hamming matchers.
 * Restates src/ORBmatcher.cc, src/MixedMatcher.cpp (type gate) and the Frame grid
 * (src/Frame.cc:431-460, 710-793) of the reference.  Everything here is reference-owned
 * arithmetic (no OpenCV numerics) except cv::BFMatcher::knnMatch semantics (orc_bf_knn2).
 */
#include "eorb_oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FRAME_GRID_ROWS 48      /* include/Frame.h:45 */
#define FRAME_GRID_COLS 64      /* include/Frame.h:46 */
#define TH_HIGH 100             /* ORBmatcher.cc:36 */
#define TH_LOW 50               /* :37 */
#define HISTO_LENGTH 30         /* :38 */

/* ORBmatcher::DescriptorDistance :2360-2378 (reads 8 x int32 whatever the descriptor width) */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* ORBmatcher::ComputeThreeMaxima :2314-2355 */
void orc_three_maxima(const int* sizes, int L, int* ind1, int* ind2, int* ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ---- Frame grid ------------------------------------------------------------------------------- */
struct orc_frame {
    int N;
    const orc_keypoint* kps;
    const uint8_t* desc; int desc_stride;
    const uint8_t* is_orb;
    orc_grid_bounds gb;
    int* cell_start;        /* COLS*ROWS+1, cell id = ix*ROWS + iy */
    int* cell_items;        /* insertion order inside each cell */
};

void orc_grid_bounds_for_image(int W, int H, orc_grid_bounds* gb)
{
    gb->minX = 0.0f; gb->maxX = (float)W; gb->minY = 0.0f; gb->maxY = (float)H;   /* Frame.cc:862-866 */
    gb->invW = (float)FRAME_GRID_COLS / (gb->maxX - gb->minX);                  /* :362-363 */
    gb->invH = (float)FRAME_GRID_ROWS / (gb->maxY - gb->minY);
}

static int frame_level(const orc_frame* f, int i)
{   /* Frame::getKPtLevelMono / MixedFrame::getKPtLevelMono (MixedFrame.cpp:438-446) */
    if (!f->is_orb || f->is_orb[i]) return f->kps[i].octave;
    return f->kps[i].class_id;
}

static int pos_in_grid(const orc_frame* f, const orc_keypoint* kp, int* px, int* py)
{   /* Frame::PosInGrid :783-793 */
    *px = (int)roundf((kp->x - f->gb.minX) * f->gb.invW);
    *py = (int)roundf((kp->y - f->gb.minY) * f->gb.invH);
    if (*px < 0 || *px >= FRAME_GRID_COLS || *py < 0 || *py >= FRAME_GRID_ROWS) return 0;
    return 1;
}

orc_frame* orc_frame_create(const orc_keypoint* kps, int N, const uint8_t* desc, int desc_stride,
                            const uint8_t* is_orb, const orc_grid_bounds* gb)
{
    orc_frame* f = (orc_frame*)calloc(1, sizeof(orc_frame));
    f->N = N; f->kps = kps; f->desc = desc; f->desc_stride = desc_stride; f->is_orb = is_orb; f->gb = *gb;
    const int ncell = FRAME_GRID_COLS * FRAME_GRID_ROWS;
    f->cell_start = (int*)calloc(ncell + 1, sizeof(int));
    f->cell_items = (int*)malloc(sizeof(int) * (N ? N : 1));
    int* cid = (int*)malloc(sizeof(int) * (N ? N : 1));
    for (int i = 0; i < N; i++) {                 /* AssignFeaturesToGrid :431-460 */
        int px, py;
        cid[i] = pos_in_grid(f, &kps[i], &px, &py) ? px * FRAME_GRID_ROWS + py : -1;
        if (cid[i] >= 0) f->cell_start[cid[i] + 1]++;
    }
    for (int c = 0; c < ncell; c++) f->cell_start[c + 1] += f->cell_start[c];
    int* fill = (int*)malloc(sizeof(int) * ncell);
    memcpy(fill, f->cell_start, sizeof(int) * ncell);
    for (int i = 0; i < N; i++) if (cid[i] >= 0) f->cell_items[fill[cid[i]]++] = i;
    free(fill); free(cid);
    return f;
}
void orc_frame_destroy(orc_frame* f) { if (f) { free(f->cell_start); free(f->cell_items); free(f); } }

/* Frame::GetFeaturesInArea :710-781 */
int orc_get_features_in_area(const orc_frame* f, float x, float y, float r, int minLevel, int maxLevel,
                             int* out, int cap)
{
    int n = 0;
    float factorX = r, factorY = r;
    int t = (int)floorf((x - f->gb.minX - factorX) * f->gb.invW);
    const int nMinCellX = t > 0 ? t : 0;
    if (nMinCellX >= FRAME_GRID_COLS) return 0;
    t = (int)ceilf((x - f->gb.minX + factorX) * f->gb.invW);
    const int nMaxCellX = t < FRAME_GRID_COLS - 1 ? t : FRAME_GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    t = (int)floorf((y - f->gb.minY - factorY) * f->gb.invH);
    const int nMinCellY = t > 0 ? t : 0;
    if (nMinCellY >= FRAME_GRID_ROWS) return 0;
    t = (int)ceilf((y - f->gb.minY + factorY) * f->gb.invH);
    const int nMaxCellY = t < FRAME_GRID_ROWS - 1 ? t : FRAME_GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * FRAME_GRID_ROWS + iy;
            for (int j = f->cell_start[c]; j < f->cell_start[c + 1]; j++) {
                const int idx = f->cell_items[j];
                if (bCheckLevels) {
                    const int level = frame_level(f, idx);
                    if (level < minLevel) continue;
                    if (maxLevel >= 0 && level > maxLevel) continue;
                }
                const float distx = f->kps[idx].x - x;
                const float disty = f->kps[idx].y - y;
                if (fabsf(distx) < factorX && fabsf(disty) < factorY) {
                    if (n < cap) out[n] = idx;
                    n++;
                }
            }
        }
    }
    return n;
}

static int rot_bin(float a1, float a2)
{   /* ORBmatcher.cc:790-796 / :2082-2087 */
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* ORBmatcher::SearchForInitialization :714-831 ; MixedMatcher.cpp:20-145 adds the isORB gate */
int orc_search_for_initialization(const orc_frame* F1, const orc_frame* F2, float* prev_matched,
                                  int* matches12, int windowSize, float nnratio, int checkOri)
{
    int nmatches = 0;
    const int N1 = F1->N, N2 = F2->N;
    for (int i = 0; i < N1; i++) matches12[i] = -1;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (N1 ? N1 : 1)); rotN[i] = 0; }
    int* matchedDist = (int*)malloc(sizeof(int) * (N2 ? N2 : 1));
    int* matches21 = (int*)malloc(sizeof(int) * (N2 ? N2 : 1));
    int* idxs = (int*)malloc(sizeof(int) * (N2 ? N2 : 1));
    for (int i = 0; i < N2; i++) { matchedDist[i] = INT_MAX; matches21[i] = -1; }
    for (int i1 = 0; i1 < N1; i1++) {
        int level1 = frame_level(F1, i1);
        if (level1 > 0) continue;
        int nc = orc_get_features_in_area(F2, prev_matched[2 * i1], prev_matched[2 * i1 + 1],
                                          (float)windowSize, level1, level1, idxs, N2);
        if (nc == 0) continue;
        const int isORB1 = !F1->is_orb || F1->is_orb[i1];
        const uint8_t* d1 = F1->desc + (size_t)i1 * F1->desc_stride;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            int i2 = idxs[c];
            const int isORB2 = !F2->is_orb || F2->is_orb[i2];
            if (isORB1 != isORB2) continue;
            int dist = orc_descriptor_distance(d1, F2->desc + (size_t)i2 * F2->desc_stride);
            if (matchedDist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if ((float)bestDist < (float)bestDist2 * nnratio) {
                if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                matches21[bestIdx2] = i1;
                matchedDist[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) {
                    int bin = rot_bin(F1->kps[i1].angle, F2->kps[bestIdx2].angle);
                    rotHist[bin][rotN[bin]++] = i1;
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) {
                int idx1 = rotHist[i][j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < N1; i1++)
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = F2->kps[matches12[i1]].x;
            prev_matched[2 * i1 + 1] = F2->kps[matches12[i1]].y;
        }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(matchedDist); free(matches21); free(idxs);
    return nmatches;
}

/* occupancy rule shared by the projection matchers (:91-93, :2045-2047):
 * a candidate holding a map point with Observations()>0 is skipped. */
static int holds_observed(const int* slot_mp, int idx, const uint8_t* mp_obs)
{
    int v = slot_mp[idx];
    if (v == -1 || v == -3) return 0;
    if (v == -2) return 1;
    return mp_obs[v] != 0;
}

/* ORBmatcher::SearchByProjection(Frame& cur, const Frame& last, th, bMono) :1969-2187, mono */
/* stereo: uright = CurrentFrame.mvuRight (NULL in the mono configurations), proj_ur = uv.x - mbf * invzc per last-frame point (:2056-2062) */
int orc_search_by_projection_last_stereo(const orc_frame* cur, const orc_frame* last, const uint8_t* valid,
                                         const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
                                         int* cur_mp, float th, int mode, int checkOri, const float* level_scale,
                                         const float* uright, const float* proj_ur);
int orc_search_by_projection_last(const orc_frame* cur, const orc_frame* last, const uint8_t* valid,
                                  const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
                                  int* cur_mp, float th, int mode, int checkOri, const float* level_scale)
{
    return orc_search_by_projection_last_stereo(cur, last, valid, uv, mp_desc, mp_obs, cur_mp, th, mode, checkOri, level_scale, NULL, NULL);
}
int orc_search_by_projection_last_stereo(const orc_frame* cur, const orc_frame* last, const uint8_t* valid,
                                         const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
                                         int* cur_mp, float th, int mode, int checkOri, const float* level_scale,
                                         const float* uright, const float* proj_ur)
{
    int nmatches = 0;
    const int Nc = cur->N;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (last->N ? last->N : 1)); rotN[i] = 0; }
    int* idxs = (int*)malloc(sizeof(int) * (Nc ? Nc : 1));
    for (int i = 0; i < last->N; i++) {
        if (!valid[i]) continue;
        const float ux = uv[2 * i], uy = uv[2 * i + 1];
        int nLastOctave = frame_level(last, i);
        float radius = th * level_scale[i];                     /* th*getORBScaleFactor(oct) (AKAZE: MixedMatcher.cpp:748-752) */
        const int isORBMP = !last->is_orb || last->is_orb[i];
        int nc;
        if (mode == 1) nc = orc_get_features_in_area(cur, ux, uy, radius, nLastOctave, -1, idxs, Nc);
        else if (mode == 2) nc = orc_get_features_in_area(cur, ux, uy, radius, 0, nLastOctave, idxs, Nc);
        else nc = orc_get_features_in_area(cur, ux, uy, radius, nLastOctave - 1, nLastOctave + 1, idxs, Nc);
        if (nc == 0) continue;
        const uint8_t* dMP = mp_desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = idxs[c];
            if (holds_observed(cur_mp, i2, mp_obs)) continue;
            if (uright && uright[i2] > 0) {                       /* :2056-2062 */
                const float ur = proj_ur[i];
                const float er = fabsf(ur - uright[i2]);
                if (er > radius) continue;
            }
            const int isORBPt = !cur->is_orb || cur->is_orb[i2];
            if (isORBMP != isORBPt) continue;                     /* MixedMatcher.cpp:787-790 */
            const int dist = orc_descriptor_distance(dMP, cur->desc + (size_t)i2 * cur->desc_stride);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_mp[bestIdx2] = i;
            nmatches++;
            if (checkOri) {
                int bin = rot_bin(last->kps[i].angle, cur->kps[bestIdx2].angle);
                rotHist[bin][rotN[bin]++] = bestIdx2;
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (int j = 0; j < rotN[i]; j++) { cur_mp[rotHist[i][j]] = -1; nmatches--; }
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(idxs);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) :44-219 mono branch;
 * MixedMatcher.cpp:500-691 adds the isORB gate (mp_is_orb / frame is_orb). */
/* stereo: uright = F.mvuRight (NULL in the mono configurations), proj_xr = pMP->mTrackProjXR per map point (:96-104) */
int orc_search_by_projection_map_stereo(const orc_frame* F, int M, const uint8_t* in_view, const float* proj_xy,
                                        const int* level, const float* view_cos, const uint8_t* mp_desc,
                                        const uint8_t* mp_obs, const uint8_t* mp_is_orb, int* frame_mp,
                                        float th, float nnratio, const float* level_scale, const float* uright, const float* proj_xr);
int orc_search_by_projection_map(const orc_frame* F, int M, const uint8_t* in_view, const float* proj_xy,
                                 const int* level, const float* view_cos, const uint8_t* mp_desc,
                                 const uint8_t* mp_obs, const uint8_t* mp_is_orb, int* frame_mp,
                                 float th, float nnratio, const float* level_scale)
{
    return orc_search_by_projection_map_stereo(F, M, in_view, proj_xy, level, view_cos, mp_desc, mp_obs, mp_is_orb, frame_mp, th, nnratio, level_scale, NULL, NULL);
}
int orc_search_by_projection_map_stereo(const orc_frame* F, int M, const uint8_t* in_view, const float* proj_xy,
                                        const int* level, const float* view_cos, const uint8_t* mp_desc,
                                        const uint8_t* mp_obs, const uint8_t* mp_is_orb, int* frame_mp,
                                        float th, float nnratio, const float* level_scale, const float* uright, const float* proj_xr)
{
    int nmatches = 0;
    const int bFactor = th != 1.0;
    int* idxs = (int*)malloc(sizeof(int) * (F->N ? F->N : 1));
    for (int m = 0; m < M; m++) {
        if (!in_view[m]) continue;
        const int nPredictedLevel = level[m];
        float r = (view_cos[m] > 0.998) ? 2.5f : 4.0f;          /* RadiusByViewingCos :221-227 */
        if (bFactor) r *= th;
        const float rs = r * level_scale[m];
        int nc = orc_get_features_in_area(F, proj_xy[2 * m], proj_xy[2 * m + 1], rs,
                                          nPredictedLevel - 1, nPredictedLevel, idxs, F->N);
        if (nc == 0) continue;
        const int isORBMP = !mp_is_orb || mp_is_orb[m];
        const uint8_t* dMP = mp_desc + 32 * (size_t)m;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = idxs[c];
            if (holds_observed(frame_mp, idx, mp_obs)) continue;
            if (uright && uright[idx] > 0) {                      /* :96-104: er > r * F.getORBScaleFactor(nPredictedLevel) */
                const float er = fabsf(proj_xr[m] - uright[idx]);
                if (er > r * level_scale[m]) continue;
            }
            const int isORBPt = !F->is_orb || F->is_orb[idx];
            if (isORBMP != isORBPt) continue;
            const int dist = orc_descriptor_distance(dMP, F->desc + (size_t)idx * F->desc_stride);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
                bestLevel = frame_level(F, idx); bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = frame_level(F, idx); bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            if (bestLevel != bestLevel2 || (float)bestDist <= nnratio * (float)bestDist2) {
                frame_mp[bestIdx] = m;
                nmatches++;
            }
        }
    }
    free(idxs);
    return nmatches;
}

/* cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, matches, 2)  (call site Frame.cc:1228).
 * OpenCV batchDistance K-NN keeps per query the K smallest distances, inserting a candidate
 * only when strictly smaller than a kept one => ties resolve to the lowest train index. */
void orc_bf_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2, int32_t* dist2)
{
    for (int i = 0; i < nq; i++) {
        int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            int d = orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
            else if (d < b1) { b1 = d; i1 = j; }
        }
        idx2[2 * i] = i0; idx2[2 * i + 1] = i1;
        dist2[2 * i] = b0; dist2[2 * i + 1] = b1;
    }
}

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>&) :276-478, mono branch */
int orc_search_by_bow(const orc_keypoint* kf_kps, int n_kf, const uint8_t* kf_desc, const uint8_t* kf_has_mp,
                      const uint32_t* kf_nodes, const int32_t* kf_node_off, const int32_t* kf_idx, int kf_nn,
                      const orc_keypoint* f_kps, int n_f, const uint8_t* f_desc,
                      const uint32_t* f_nodes, const int32_t* f_node_off, const int32_t* f_idx, int f_nn,
                      int32_t* match_f, float nnratio, int checkOri)
{
    (void)n_kf;
    int nmatches = 0;
    for (int i = 0; i < n_f; i++) match_f[i] = -1;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n_f ? n_f : 1)); rotN[i] = 0; }
    int a = 0, b = 0;
    while (a < kf_nn && b < f_nn) {
        if (kf_nodes[a] == f_nodes[b]) {
            for (int iKF = kf_node_off[a]; iKF < kf_node_off[a + 1]; iKF++) {
                const int realIdxKF = kf_idx[iKF];
                if (!kf_has_mp[realIdxKF]) continue;                          /* !pMP || pMP->isBad() */
                const uint8_t* dKF = kf_desc + 32 * (size_t)realIdxKF;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int iF = f_node_off[b]; iF < f_node_off[b + 1]; iF++) {
                    const int realIdxF = f_idx[iF];
                    if (match_f[realIdxF] >= 0) continue;                     /* vpMapPointMatches[realIdxF] */
                    const int dist = orc_descriptor_distance(dKF, f_desc + 32 * (size_t)realIdxF);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        match_f[bestIdxF] = realIdxKF;
                        if (checkOri) {
                            int bin = rot_bin(kf_kps[realIdxKF].angle, f_kps[bestIdxF].angle);
                            rotHist[bin][rotN[bin]++] = bestIdxF;
                        }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (kf_nodes[a] < f_nodes[b]) {
            while (a < kf_nn && kf_nodes[a] < f_nodes[b]) a++;                /* lower_bound */
        } else {
            while (b < f_nn && f_nodes[b] < kf_nodes[a]) b++;
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) { match_f[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    return nmatches;
}

/* MixedFrame::sortFeaturesResponse: multimap<float, ..., greater<float>> keeps insertion order among equal keys */
void orc_sort_by_response(const orc_keypoint* kps, int n, int32_t* perm)
{
    for (int i = 0; i < n; i++) {           /* stable insertion into a descending sequence */
        int k = i;
        while (k > 0 && kps[perm[k - 1]].response < kps[i].response) { perm[k] = perm[k - 1]; k--; }
        perm[k] = i;
    }
}

/* MixedFrame::resolveNumMixedPts (MixedFrame.cpp:281-317) */
void orc_resolve_num_mixed(int nDetectedORB, int nDetectedAK, int nDesired, int nDesiredAK, int* nORB, int* nAK)
{
    const int nDetected = nDetectedORB + nDetectedAK;
    const int nDesiredORB = nDesired - nDesiredAK;
    if (nDetected > nDesired) {
        const int nDiff = nDetected - nDesired;
        if (nDetectedORB > nDesiredORB && nDetectedAK > nDesiredAK) { *nORB = nDesiredORB; *nAK = nDesiredAK; }
        else if (nDetectedORB > nDesiredORB) { *nORB = nDetectedORB - nDiff; *nAK = nDetectedAK < nDesiredAK ? nDetectedAK : nDesiredAK; }
        else if (nDetectedAK > nDesiredAK) { *nORB = nDetectedORB < nDesiredORB ? nDetectedORB : nDesiredORB; *nAK = nDetectedAK - nDiff; }
        else { /* the reference only prints an error and leaves the counts untouched */ }
    } else { *nORB = nDetectedORB; *nAK = nDetectedAK; }
}

/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) :833-973, mono */
int orc_search_by_bow_kf(const orc_keypoint* kps1, int n1, const uint8_t* desc1, const uint8_t* has_mp1,
                         const uint32_t* nodes1, const int32_t* off1, const int32_t* idx1, int nn1,
                         const orc_keypoint* kps2, int n2, const uint8_t* desc2, const uint8_t* has_mp2,
                         const uint32_t* nodes2, const int32_t* off2, const int32_t* idx2, int nn2,
                         int32_t* match12, float nnratio, int checkOri)
{
    int nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    uint8_t* matched2 = (uint8_t*)calloc(n2 ? n2 : 1, 1);
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n1 ? n1 : 1)); rotN[i] = 0; }
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            for (int i1 = off1[a]; i1 < off1[a + 1]; i1++) {
                const int id1 = idx1[i1];
                if (!has_mp1[id1]) continue;
                const uint8_t* d1 = desc1 + 32 * (size_t)id1;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                    const int id2 = idx2[i2];
                    if (matched2[id2] || !has_mp2[id2]) continue;
                    const int dist = orc_descriptor_distance(d1, desc2 + 32 * (size_t)id2);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = id2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        match12[id1] = bestIdx2;
                        matched2[bestIdx2] = 1;
                        if (checkOri) {
                            int bin = rot_bin(kps1[id1].angle, kps2[bestIdx2].angle);
                            rotHist[bin][rotN[bin]++] = id1;
                        }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) a++;
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) b++;
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(matched2);
    return nmatches;
}

static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

/* MapPoint::ComputeDistinctiveDescriptors src/MapPoint.cc:349-423 */
void orc_distinctive_descriptors(const uint8_t* desc, const int32_t* offsets, int M, int32_t* best)
{
    for (int m = 0; m < M; m++) {
        const int N = offsets[m + 1] - offsets[m];
        if (N <= 0) { best[m] = -1; continue; }
        const uint8_t* D = desc + 32 * (size_t)offsets[m];
        int* row = (int*)malloc(sizeof(int) * N);
        int BestMedian = INT_MAX, BestIdx = 0;
        for (int i = 0; i < N; i++) {
            for (int j = 0; j < N; j++) row[j] = (i == j) ? 0 : orc_descriptor_distance(D + 32 * (size_t)i, D + 32 * (size_t)j);
            qsort(row, N, sizeof(int), cmp_int);
            const int median = row[(int)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        free(row);
        best[m] = BestIdx;
    }
}

/* Pinhole::epipolarConstrain (src/CameraModels/Pinhole.cpp:134-157) with F12 given */
static int epipolar_ok(const orc_keypoint* kp1, const orc_keypoint* kp2, const float* F, float unc)
{
    const float a = kp1->x * F[0] + kp1->y * F[3] + F[6];
    const float b = kp1->x * F[1] + kp1->y * F[4] + F[7];
    const float c = kp1->x * F[2] + kp1->y * F[5] + F[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84f * unc;                       /* DEF_EC_DIST_COEF, include/CameraModels/Pinhole.h:36 */
}

/* ORBmatcher::SearchForTriangulation :975-1214 (mono: no mpCamera2, mvuRight < 0, bOnlyStereo = false) */
int orc_search_for_triangulation(const orc_keypoint* kps1, int n1, const uint8_t* desc1, int stride1, const uint8_t* elig1,
                                 const uint32_t* nodes1, const int32_t* off1, const int32_t* idx1, int nn1,
                                 const orc_keypoint* kps2, int n2, const uint8_t* desc2, int stride2, const uint8_t* elig2,
                                 const uint32_t* nodes2, const int32_t* off2, const int32_t* idx2, int nn2,
                                 const float ep[2], const float F12[9], const float* scale2, const float* sigma2_2,
                                 int bCoarse, int checkOri, int32_t* match12)
{
    (void)n2;
    int nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n1 ? n1 : 1)); rotN[i] = 0; }
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            for (int i1 = off1[a]; i1 < off1[a + 1]; i1++) {
                const int id1 = idx1[i1];
                if (!(elig1[id1] & 1)) continue;                             /* pMP1 (:1046) / !isORBDescValid; bit 1: bStereo1 (:1051) */
                const orc_keypoint* kp1 = &kps1[id1];
                const uint8_t* d1 = desc1 + (size_t)stride1 * id1;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                    const int id2 = idx2[i2];
                    if (!(elig2[id2] & 1)) continue;                         /* vbMatched2 is never set in this function */
                    const int dist = orc_descriptor_distance(d1, desc2 + (size_t)stride2 * id2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const orc_keypoint* kp2 = &kps2[id2];
                    if (!((elig1[id1] | elig2[id2]) & 2)) {                  /* if(!bStereo1 && !bStereo2 && !pKF1->mpCamera2) :1093 */
                        const float distex = ep[0] - kp2->x, distey = ep[1] - kp2->y;
                        if (distex * distex + distey * distey < 100 * scale2[kp2->octave]) continue;
                    }
                    if (epipolar_ok(kp1, kp2, F12, sigma2_2[kp2->octave]) || bCoarse) { bestIdx2 = id2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    match12[id1] = bestIdx2;
                    nmatches++;
                    if (checkOri) {
                        int bin = rot_bin(kp1->angle, kps2[bestIdx2].angle);
                        rotHist[bin][rotN[bin]++] = id1;
                    }
                }
            }
            a++; b++;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) a++;
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) b++;
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    return nmatches;
}

/* Fuse :1512-1578 / Fuse(Scw) :1619-1741 / SearchBySim3 :1829-1860 / SearchByProjection(KF,Scw) :548-588 search core */
static void kf_radius_core(const orc_frame* kf, int M, const uint8_t* valid, const float* uv, const float* radius,
                         const int32_t* level, const uint8_t* q_desc, const float* inv_sigma2, uint8_t* taken,
                         float accept_thr, int32_t* best_idx, int32_t* best_dist, const float* uright, const float* q_ur)
{
    int* cand = (int*)malloc(sizeof(int) * (kf->N ? kf->N : 1));
    for (int m = 0; m < M; m++) {
        best_idx[m] = -1; best_dist[m] = 256;
        if (!valid[m]) continue;
        const float u = uv[2 * m], v = uv[2 * m + 1];
        const int nc = orc_get_features_in_area(kf, u, v, radius[m], -1, -1, cand, kf->N);
        const int L = level[m];
        int bestDist = 256, bestIdx = -1;
        for (int j = 0; j < nc; j++) {
            const int idx = cand[j];
            if (taken && taken[idx]) continue;
            const int kpLevel = kf->kps[idx].octave;
            if (kpLevel < L - 1 || kpLevel > L) continue;
            if (inv_sigma2) {
                const float ex = u - kf->kps[idx].x, ey = v - kf->kps[idx].y;
                if (uright && uright[idx] >= 0) {                            /* "Check reprojection error in stereo" :1541-1553 */
                    const float er = q_ur[m] - uright[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = orc_descriptor_distance(q_desc + 32 * (size_t)m, kf->desc + (size_t)kf->desc_stride * idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[m] = bestIdx; best_dist[m] = bestDist;
        if (taken && bestIdx >= 0 && (float)bestDist <= accept_thr) taken[bestIdx] = 1;
    }
    free(cand);
}

void orc_kf_radius_match(const orc_frame* kf, int M, const uint8_t* valid, const float* uv, const float* radius,
                         const int32_t* level, const uint8_t* q_desc, const float* inv_sigma2, uint8_t* taken,
                         float accept_thr, int32_t* best_idx, int32_t* best_dist)
{
    kf_radius_core(kf, M, valid, uv, radius, level, q_desc, inv_sigma2, taken, accept_thr, best_idx, best_dist, NULL, NULL);
}

/* Fuse on a rectified-stereo KeyFrame: uright = pKF->mvuRight, q_ur[m] = u - bf*invz (:1506, :1541-1553) */
void orc_kf_radius_match_stereo(const orc_frame* kf, int M, const uint8_t* valid, const float* uv, const float* radius,
                         const int32_t* level, const uint8_t* q_desc, const float* inv_sigma2,
                         const float* uright, const float* q_ur, int32_t* best_idx, int32_t* best_dist)
{
    kf_radius_core(kf, M, valid, uv, radius, level, q_desc, inv_sigma2, NULL, 0.f, best_idx, best_dist, uright, q_ur);
}

/* ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) :2189-2312 / MixedMatcher.cpp:928-1063 */
int orc_search_by_projection_kf(const orc_frame* cur, const orc_keypoint* kf_kps, int n_kf, const uint8_t* kf_is_orb,
                                const uint8_t* valid, const float* uv, const int32_t* pred_level, const float* level_scale,
                                const uint8_t* mp_desc, int* cur_mp, float th, int ORBdist, int checkOri)
{
    int nmatches = 0;
    const int Nc = cur->N;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n_kf ? n_kf : 1)); rotN[i] = 0; }
    int* idxs = (int*)malloc(sizeof(int) * (Nc ? Nc : 1));
    for (int i = 0; i < n_kf; i++) {
        if (!valid[i]) continue;
        const int nPredictedLevel = pred_level[i];
        const float radius = th * level_scale[i];
        const int nc = orc_get_features_in_area(cur, uv[2 * i], uv[2 * i + 1], radius, nPredictedLevel - 1, nPredictedLevel + 1, idxs, Nc);
        if (nc == 0) continue;
        const int isORBMP = !kf_is_orb || kf_is_orb[i];
        const uint8_t* dMP = mp_desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = idxs[c];
            if (cur_mp[i2] != -1) continue;                           /* CurrentFrame.getMapPoint(i2) */
            const int isORBPt = !cur->is_orb || cur->is_orb[i2];
            if (isORBMP != isORBPt) continue;
            const int dist = orc_descriptor_distance(dMP, cur->desc + (size_t)i2 * cur->desc_stride);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {
            cur_mp[bestIdx2] = i;
            nmatches++;
            if (checkOri) {
                int bin = rot_bin(kf_kps[i].angle, cur->kps[bestIdx2].angle);
                rotHist[bin][rotN[bin]++] = bestIdx2;
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (int j = 0; j < rotN[i]; j++) { cur_mp[rotHist[i][j]] = -1; nmatches--; }
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(idxs);
    return nmatches;
}

/* FORB::distance, Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101 (bit-parallel popcount over 8 words) */
static int forb_distance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb; memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

typedef struct { uint32_t key; int idx; double w; } bow_item;
static int cmp_bow_item(const void* a, const void* b)
{
    const bow_item* x = (const bow_item*)a; const bow_item* y = (const bow_item*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx - y->idx;
}

/* TemplatedVocabulary::transform :1125-1190 and :1208-1250 */
void orc_bow_transform(const orc_vocabulary* voc, const uint8_t* desc, int n, int stride, int levelsup, int weighting, int norm,
                       uint32_t* bow_word, double* bow_val, int* n_words, uint32_t* fv_node, int32_t* fv_off, int32_t* fv_idx,
                       int* n_fvnodes, int32_t* word_of, int32_t* node_of)
{
    *n_words = 0; *n_fvnodes = 0; fv_off[0] = 0;
    if (voc->nnodes <= 1 || n <= 0) return;                              /* empty() :1132 */
    bow_item* words = (bow_item*)malloc(sizeof(bow_item) * n);
    bow_item* nodes = (bow_item*)malloc(sizeof(bow_item) * n);
    int m = 0;
    const int nid_level = voc->L - levelsup;
    for (int f = 0; f < n; f++) {
        const uint8_t* feature = desc + (size_t)stride * f;
        int nid = 0;                                                      /* if(nid_level <= 0) *nid = 0 */
        int final_id = 0, current_level = 0;
        do {
            ++current_level;
            const int c0 = voc->child_off[final_id], c1 = voc->child_off[final_id + 1];
            final_id = voc->child_ids[c0];
            int best_d = forb_distance(feature, voc->node_desc + 32 * (size_t)final_id);
            for (int c = c0 + 1; c < c1; c++) {
                const int id = voc->child_ids[c];
                const int d = forb_distance(feature, voc->node_desc + 32 * (size_t)id);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (voc->child_off[final_id + 1] > voc->child_off[final_id]);
        const double w = voc->weight[final_id];
        if (word_of) word_of[f] = w > 0 ? voc->word_id[final_id] : -1;
        if (node_of) node_of[f] = w > 0 ? nid : -1;
        if (w > 0) {                                                      /* not stopped */
            words[m].key = (uint32_t)voc->word_id[final_id]; words[m].idx = f; words[m].w = w;
            nodes[m].key = (uint32_t)nid; nodes[m].idx = f; nodes[m].w = 0;
            m++;
        }
    }
    qsort(words, m, sizeof(bow_item), cmp_bow_item);
    qsort(nodes, m, sizeof(bow_item), cmp_bow_item);
    /* BowVector: addWeight accumulates in feature order (TF / TF_IDF), addIfNotExist keeps the first (IDF / BINARY) */
    int nw = 0;
    for (int i = 0; i < m; ) {
        int j = i; double v = words[i].w;
        for (j = i + 1; j < m && words[j].key == words[i].key; j++)
            if (weighting == 0 || weighting == 1) v += words[j].w;
        bow_word[nw] = words[i].key; bow_val[nw] = v; nw++;
        i = j;
    }
    if ((weighting == 0 || weighting == 1) && nw > 0 && norm == 0) {
        const double nd = (double)nw;
        for (int i = 0; i < nw; i++) bow_val[i] /= nd;
    }
    if (norm != 0) {                                                      /* BowVector::normalize, BowVector.cpp:57-79 */
        double nrm = 0.0;
        if (norm == 1) { for (int i = 0; i < nw; i++) nrm += fabs(bow_val[i]); }
        else { for (int i = 0; i < nw; i++) nrm += bow_val[i] * bow_val[i]; nrm = sqrt(nrm); }
        if (nrm > 0.0) for (int i = 0; i < nw; i++) bow_val[i] /= nrm;
    }
    *n_words = nw;
    int nn = 0;
    for (int i = 0; i < m; ) {
        int j = i;
        fv_node[nn] = nodes[i].key;
        for (; j < m && nodes[j].key == nodes[i].key; j++) fv_idx[j] = nodes[j].idx;
        nn++; fv_off[nn] = j;
        i = j;
    }
    *n_fvnodes = nn;
    free(words); free(nodes);
}

/* shared inner loop of the windowed matchers (ORBmatcher.cc:754-774 and its siblings) */
void orc_hamming_window_match(const uint8_t* q_desc, int nq, int q_stride, const uint8_t* t_desc, int t_stride,
                              const int32_t* cand_offsets, const int32_t* cand_idx, int32_t* best_idx, int32_t* best_d,
                              int32_t* second_idx, int32_t* second_d)
{
    for (int q = 0; q < nq; q++) {
        int bd = 256, bi = -1, sd = 256, si = -1;
        for (int k = cand_offsets[q]; k < cand_offsets[q + 1]; k++) {
            const int c = cand_idx[k];
            const int d = orc_descriptor_distance(q_desc + (size_t)q_stride * q, t_desc + (size_t)t_stride * c);
            if (d < bd) { sd = bd; si = bi; bd = d; bi = c; }
            else if (d < sd) { sd = d; si = c; }
        }
        best_idx[q] = bi; best_d[q] = bd; second_idx[q] = si; second_d[q] = sd;
    }
}
