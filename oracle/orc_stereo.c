/* orc_stereo.c -- CPU restatement of Frame::ComputeStereoMatches (src/Frame.cc:869-1048 of the reference), rectified stereo.
 * TEST INFRASTRUCTURE (see eorb_oracle.h): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * Parity unpinned, like the rest of oracle/: the reference holds no fixture for this path.
 *
 * The left and the right image went through two extractors with equal parameters (src/Frame.cc:122-125: ExtractORB(0, imLeft, 0, 0) /
 * ExtractORB(1, imRight, 0, 0)); their level images (mvImagePyramid[level], the un-blurred resized levels) are what the
 * sliding-window correlation reads.  Restated line by line:
 *   :880-899   the row table: right keypoint iR is a candidate of every image row in [floor(y - r), ceil(y + r)], r = 2 * scale(octave)
 *   :909-959   per left keypoint: the row vRowIndices[(size_t)vL], octave within +-1, uR in [uL - maxD, uL], smallest descriptor
 *              distance below TH_HIGH, the first one in iR order among equals (`dist < bestDist`)
 *   :962-1033  bestDist < (TH_HIGH + TH_LOW) / 2: L1 norm of the 11 x 11 patches at the keypoint's level over the shifts -5..5 (the
 *              first smallest), parabola through the three values around it, the re-scaled right coordinate, disparity and depth
 *   :1036-1047 the matches sorted by (L1 norm, iL); everything at or above 1.5 * 1.4 * the median norm is taken back
 * cv::norm(IL, IR, NORM_L1) of 8-bit patches is the integer sum of absolute differences (exact in double and in float: <= 30 855).
 * Deviations that cannot occur with keypoints of the extractor (>= EDGE_THRESHOLD - 3 = 16 pixels inside their level): a patch that
 * would leave the level image makes OpenCV throw; here the keypoint is skipped.  An empty match list makes the reference read
 * vDistIdx[0] of an empty vector; here nothing is taken back. */
#include "eorb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

typedef struct { int dist, iL; } dist_idx;
static int cmp_dist_idx(const void* a, const void* b)
{
    const dist_idx* p = (const dist_idx*)a; const dist_idx* q = (const dist_idx*)b;
    if (p->dist != q->dist) return p->dist < q->dist ? -1 : 1;
    return p->iL < q->iL ? -1 : (p->iL > q->iL ? 1 : 0);
}

int orc_compute_stereo_matches(const orc_orb* eL, const orc_orb* eR, const orc_keypoint* kL, int N, const uint8_t* dL,
                               const orc_keypoint* kR, int Nr, const uint8_t* dR, float mb, float mbf, float* uRight, float* depth)
{
    const int TH_HIGH = 100, TH_LOW = 50;                                  /* src/ORBmatcher.cc:36-37 */
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const float* sf = orc_orb_scale_factors(eL);
    const float* inv_sf = orc_orb_inv_scale_factors(eL);
    int W0, nRows;
    if (orc_orb_level_size(eL, 0, &W0, &nRows)) return -1;
    for (int i = 0; i < N; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    /* row table :880-899 (push_back order = iR order) */
    int* cnt = (int*)calloc((size_t)nRows + 1, sizeof(int));
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * sf[kR[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) cnt[yi]++;
    }
    int* off = (int*)malloc(((size_t)nRows + 1) * sizeof(int));
    off[0] = 0;
    for (int y = 0; y < nRows; y++) off[y + 1] = off[y] + cnt[y];
    int* rows = (int*)malloc((size_t)(off[nRows] > 0 ? off[nRows] : 1) * sizeof(int));
    memset(cnt, 0, ((size_t)nRows + 1) * sizeof(int));
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * sf[kR[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) rows[off[yi] + cnt[yi]++] = iR;
    }
    const float minZ = mb, minD = 0.f, maxD = mbf / minZ;
    dist_idx* vDistIdx = (dist_idx*)malloc((size_t)(N > 0 ? N : 1) * sizeof(dist_idx));
    int nd = 0;
    for (int iL = 0; iL < N; iL++) {
        const int levelL = kL[iL].octave;
        const float vL = kL[iL].y, uL = kL[iL].x;
        const long rowi = (long)vL;                                        /* vRowIndices[vL]: float -> size_t */
        if (rowi < 0 || rowi >= nRows) continue;
        const int c0 = off[rowi], c1 = off[rowi + 1];
        if (c0 == c1) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH; int bestIdxR = 0;
        for (int iC = c0; iC < c1; iC++) {
            const int iR = rows[iC];
            if (kR[iR].octave < levelL - 1 || kR[iR].octave > levelL + 1) continue;
            const float uR = kR[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(dL + 32 * (size_t)iL, dR + 32 * (size_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist >= thOrbDist) continue;
        /* sub-pixel match by correlation :962-1033 */
        const float uR0 = kR[bestIdxR].x;
        const float scaleFactor = inv_sf[levelL];
        const float scaleduL = roundf(uL * scaleFactor), scaledvL = roundf(vL * scaleFactor), scaleduR0 = roundf(uR0 * scaleFactor);
        const int w = 5, L = 5;
        int lw, lh, bwL, bhL, bwR, bhR;
        orc_orb_level_size(eL, levelL, &lw, &lh);
        const uint8_t* bufL = orc_orb_level_buffer(eL, levelL, &bwL, &bhL);
        const uint8_t* bufR = orc_orb_level_buffer(eR, levelL, &bwR, &bhR);
        const int E = orc_orb_edge_threshold(eL);
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= (float)lw) continue;
        const int r0 = (int)(scaledvL - w), cL0 = (int)(scaleduL - w);
        /* (patches inside the level images: always true for the extractor's keypoints, see the header) */
        if (r0 < 0 || r0 + 2 * w + 1 > lh || cL0 < 0 || cL0 + 2 * w + 1 > lw) continue;
        if ((int)(scaleduR0 - L - w) < 0 || (int)(scaleduR0 + L + w + 1) > lw) continue;
        int bestD = INT_MAX, bestincR = 0;
        float vDists[11];
        for (int incR = -L; incR <= L; incR++) {
            const int cR0 = (int)(scaleduR0 + incR - w);
            int s = 0;
            for (int y = 0; y < 2 * w + 1; y++) {
                const uint8_t* pl = bufL + (size_t)(r0 + y + E) * bwL + (cL0 + E);
                const uint8_t* pr = bufR + (size_t)(r0 + y + E) * bwR + (cR0 + E);
                for (int x = 0; x < 2 * w + 1; x++) s += abs((int)pl[x] - (int)pr[x]);
            }
            const float dist = (float)s;
            if (dist < (float)bestD) { bestD = (int)dist; bestincR = incR; }
            vDists[L + incR] = dist;
        }
        if (bestincR == -L || bestincR == L) continue;
        const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = sf[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
        float disparity = uL - bestuR;
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }      /* (double literals: uL - 0.01 is a double difference) */
            depth[iL] = mbf / disparity;
            uRight[iL] = bestuR;
            vDistIdx[nd].dist = bestD; vDistIdx[nd].iL = iL; nd++;
        }
    }
    if (nd > 0) {
        qsort(vDistIdx, (size_t)nd, sizeof(dist_idx), cmp_dist_idx);
        const float median = (float)vDistIdx[nd / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nd - 1; i >= 0; i--) {
            if ((float)vDistIdx[i].dist < thDist) break;
            uRight[vDistIdx[i].iL] = -1; depth[vDistIdx[i].iL] = -1;
        }
    }
    free(vDistIdx); free(rows); free(off); free(cnt);
    return nd;
}
