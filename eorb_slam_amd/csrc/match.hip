// match.hip -- 256-bit Hamming matchers on gfx950 (wavefront __popcll), bit-exact w.r.t. the reference's
// sequential matchers.
//
//   bf_knn2_kernel        brute-force 2-NN (cv::BFMatcher::knnMatch(.,.,2) at src/Frame.cc:1228):
//                         16 lanes share one query and stride over an LDS-staged train tile; the two
//                         smallest (dist, trainIdx) keys are merged with shuffles.
//   search_init_kernel    ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:714-831) /
//                         MixedMatcher (src/MixedMatcher.cpp:20-145).  The reference is a greedy
//                         sequential loop whose state (vMatchedDistance, vnMatches21) feeds later
//                         queries, so queries stay sequential; the candidate scan of each query
//                         (Frame::GetFeaturesInArea, src/Frame.cc:710-781 + DescriptorDistance,
//                         ORBmatcher.cc:2360-2378) is spread over the 256 threads of a workgroup.
//                         Tie-breaks follow the reference's candidate order (cell ix, cell iy,
//                         insertion index) through a composed sort key.
//   search_proj_*_kernel  the two tracking SearchByProjection variants (ORBmatcher.cc:44-219,
//                         :1969-2187), same structure.
// One workgroup per frame pair; batches of pairs run concurrently.
#include "eorb_ctx.h"
#include "match_args.h"
#include "dev_math.h"
#include <algorithm>

namespace eorb {

constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;      // ORBmatcher.cc:36-38

// ---------------------------------------------------------------------------------------------------
// brute force 2-NN.  LPQ lanes share a query (16, 32 or 64: 16 / 8 / 4 queries per workgroup); the launcher picks the widest
// sharing that still gives every CU a workgroup (2 000 queries: 125 workgroups at 16 lanes per query, 500 at 64)
template <int LPQ>
__global__ __launch_bounds__(256) void bf_knn2_kernel(const uint8_t* __restrict__ q, int nq,
                                                      const uint8_t* __restrict__ t, int nt,
                                                      int32_t* __restrict__ idx2, int32_t* __restrict__ dist2)
{
    __shared__ uint64_t tile[256 * 4];
    const int tid = threadIdx.x;
    const int part = tid & (LPQ - 1);
    const int qi = blockIdx.x * (256 / LPQ) + tid / LPQ;
    uint64_t qa[4] = {0, 0, 0, 0};
    if (qi < nq) {
        const uint64_t* qp = (const uint64_t*)(q + (size_t)qi * 32);
        qa[0] = qp[0]; qa[1] = qp[1]; qa[2] = qp[2]; qa[3] = qp[3];
    }
    // keys: dist << 32 | trainIdx ; smaller key = better; ties -> lowest train index
    uint64_t k0 = ~0ull, k1 = ~0ull;
    for (int t0 = 0; t0 < nt; t0 += 256) {
        const int nload = min(256, nt - t0);
        __syncthreads();
        if (tid < nload) {
            const uint64_t* tp = (const uint64_t*)(t + (size_t)(t0 + tid) * 32);
            tile[tid * 4 + 0] = tp[0]; tile[tid * 4 + 1] = tp[1]; tile[tid * 4 + 2] = tp[2]; tile[tid * 4 + 3] = tp[3];
        }
        __syncthreads();
        for (int j = part; j < nload; j += LPQ) {
            const uint64_t* tp = &tile[j * 4];
            const int d = __popcll(qa[0] ^ tp[0]) + __popcll(qa[1] ^ tp[1]) + __popcll(qa[2] ^ tp[2]) + __popcll(qa[3] ^ tp[3]);
            const uint64_t key = ((uint64_t)d << 32) | (uint32_t)(t0 + j);
            if (key < k0) { k1 = k0; k0 = key; }
            else if (key < k1) k1 = key;
        }
    }
    // merge the LPQ partial top-2 lists of a query
#pragma unroll
    for (int d = LPQ / 2; d >= 1; d >>= 1) {
        const uint64_t o0 = __shfl_xor(k0, d, 64), o1 = __shfl_xor(k1, d, 64);
        // top-2 of {k0,k1,o0,o1}
        const uint64_t lo = k0 < o0 ? k0 : o0;
        const uint64_t hi = k0 < o0 ? o0 : k0;
        const uint64_t s = k1 < o1 ? k1 : o1;
        k0 = lo; k1 = hi < s ? hi : s;
    }
    if (part == 0 && qi < nq) {
        idx2[2 * qi] = (k0 == ~0ull) ? -1 : (int32_t)(k0 & 0xffffffffu);
        dist2[2 * qi] = (k0 == ~0ull) ? 0x7fffffff : (int32_t)(k0 >> 32);
        idx2[2 * qi + 1] = (k1 == ~0ull) ? -1 : (int32_t)(k1 & 0xffffffffu);
        dist2[2 * qi + 1] = (k1 == ~0ull) ? 0x7fffffff : (int32_t)(k1 >> 32);
    }
}


// 32 descriptor bytes -> 4 x u64 (little endian).  AKAZE rows are 61 B apart (src/MixedFrame.cpp:20-21), so a
// row may start at any byte: assemble bytewise unless 8-byte aligned.
__device__ __forceinline__ void load_desc32(const uint8_t* __restrict__ p, uint64_t& a, uint64_t& b, uint64_t& c, uint64_t& d)
{
    if ((((uintptr_t)p) & 7) == 0) {
        const uint64_t* q = (const uint64_t*)p;
        a = q[0]; b = q[1]; c = q[2]; d = q[3];
    } else {
        uint64_t v[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint64_t x = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) x |= (uint64_t)p[w * 8 + k] << (8 * k);
            v[w] = x;
        }
        a = v[0]; b = v[1]; c = v[2]; d = v[3];
    }
}

// ---------------------------------------------------------------------------------------------------
// shared pieces of the greedy window matchers

__device__ __forceinline__ int rot_bin(float a1, float a2)
{   // ORBmatcher.cc:790-796
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// Frame::GetFeaturesInArea cell range (src/Frame.cc:722-745); returns false when the reference returns empty
__device__ __forceinline__ bool cell_range(const GridB& g, float x, float y, float r, int& cx0, int& cx1, int& cy0, int& cy1)
{
    cx0 = max(0, (int)floorf((x - g.minX - r) * g.invW));
    if (cx0 >= kGridCols) return false;
    cx1 = min(kGridCols - 1, (int)ceilf((x - g.minX + r) * g.invW));
    if (cx1 < 0) return false;
    cy0 = max(0, (int)floorf((y - g.minY - r) * g.invH));
    if (cy0 >= kGridRows) return false;
    cy1 = min(kGridRows - 1, (int)ceilf((y - g.minY + r) * g.invH));
    if (cy1 < 0) return false;
    return true;
}

// block-wide top-2 of 64-bit keys (smaller = better). Result valid in all threads.
__device__ __forceinline__ void block_top2(uint64_t& k0, uint64_t& k1, uint64_t* red /* 2*nwaves */)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t o0 = __shfl_xor(k0, d, 64), o1 = __shfl_xor(k1, d, 64);
        const uint64_t lo = k0 < o0 ? k0 : o0;
        const uint64_t hi = k0 < o0 ? o0 : k0;
        const uint64_t s = k1 < o1 ? k1 : o1;
        k0 = lo; k1 = hi < s ? hi : s;
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[2 * w] = k0; red[2 * w + 1] = k1; }
    __syncthreads();
    uint64_t a0 = red[0], a1 = red[1];
    for (int i = 1; i < nw; i++) {
        const uint64_t o0 = red[2 * i], o1 = red[2 * i + 1];
        const uint64_t lo = a0 < o0 ? a0 : o0;
        const uint64_t hi = a0 < o0 ? o0 : a0;
        const uint64_t s = a1 < o1 ? a1 : o1;
        a0 = lo; a1 = hi < s ? hi : s;
    }
    k0 = a0; k1 = a1;
}

// ComputeThreeMaxima (ORBmatcher.cc:2314-2355)
// (selects on locals: the if / else-if ladder over reference arguments was compiled to 150 scratch accesses per thread)
__device__ __forceinline__ void three_maxima(const int* histo, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    int a = -1, b = -1, c = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        const bool g1 = s > max1, g2 = s > max2, g3 = s > max3;      // (max1 >= max2 >= max3: g1 implies g2 implies g3)
        c = g2 ? b : (g3 ? i : c); max3 = g2 ? max2 : (g3 ? s : max3);
        b = g1 ? a : (g2 ? i : b); max2 = g1 ? max1 : (g2 ? s : max2);
        a = g1 ? i : a;            max1 = g1 ? s : max1;
    }
    if ((float)max2 < 0.1f * (float)max1) { b = -1; c = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { c = -1; }
    ind1 = a; ind2 = b; ind3 = c;
}

__device__ __forceinline__ int kp_level(const eorb_keypoint& k, bool isorb)
{   // Frame::getKPtLevelMono / MixedFrame::getKPtLevelMono (MixedFrame.cpp:438-446)
    return isorb ? k.octave : k.class_id;
}

// ---------------------------------------------------------------------------------------------------
// The greedy window matchers: ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:714-831, MixedMatcher.cpp:20-145), the two
// tracking SearchByProjection variants (ORBmatcher.cc:44-219, :1969-2187; MixedMatcher.cpp:500-926) and the relocalisation variant
// (:2189-2312), mono branches.  The reference walks the queries in order because a query's outcome depends on what earlier queries
// matched (vMatchedDistance / vnMatches21; setMapPoint -> "holds an observed map point").  That state only FILTERS candidates; the
// expensive part -- Frame::GetFeaturesInArea (src/Frame.cc:710-781) + DescriptorDistance (ORBmatcher.cc:2360-2378) over the window
// -- does not depend on it.  Two phases:
//   win_cand_kernel     every (query, candidate) pair in parallel: one wavefront per query, lanes over the searched frame (staged in
//                       LDS), window / level / type tests, distance; the candidates whose distance can still influence the outcome
//                       (dist < dmax, see the launchers) form the query's list, sorted by key = dist | cell | index | level: the
//                       reference's candidate order (cell ix, cell iy, insertion index) rides in the key for the tie-breaks.
//   win_resolve_kernel  one wavefront per frame pair walks the queries in order: the (short) list of a query is filtered by the
//                       current state, the two smallest keys found by a wave reduction, the greedy update applied.  Everything it
//                       touches sits in LDS; the rotation histogram (ComputeThreeMaxima, :2314-2355) is filled afterwards in
//                       parallel from the recorded (query, match) pairs (a count is order-independent).
// A query whose list outgrows its capacity (or the pair's pool) is resolved by a full scan of the searched frame inside phase 2.
constexpr uint32_t kWinOver = 0xFFFFFFFFu;        // cnt value: list overflowed, resolve by a full scan

struct WinArgs {
    // searched frame (per pair: pointer + pair * stride)
    const eorb_keypoint* kps2; size_t kp2_stride; const int32_t* n2p; int n2; const uint8_t* desc2; int dstride2; size_t desc2_slice;
    const uint8_t* is_orb2; int cap2;
    // queries
    const int32_t* nqp; int nq; int capq;
    // SearchForInitialization: queries = keypoints of frame 1
    const eorb_keypoint* kps1; size_t kp1_stride; const uint8_t* desc1; int dstride1; size_t desc1_slice; const uint8_t* is_orb1;
    float* prev_matched; int windowSize;
    // SearchByProjection: queries = last-frame keypoints / map points
    const eorb_keypoint* qkps; const uint8_t* q_is_orb; const uint8_t* valid; const float* qf; const int32_t* qlevel;
    const uint8_t* mp_desc; const uint8_t* mp_obs; const uint8_t* mp_is_orb; float th; int mode; int dist_th;
    // stereo gate (src/ORBmatcher.cc:96-104, :2056-2062): uright2 = mvuRight of the searched frame's keypoints (> 0: has a right match),
    // q_ur = the query's right coordinate (mTrackProjXR, or uv.x - mbf * invzc); both NULL in the mono configurations
    const float* uright2; const float* q_ur;
    GridB g; float nnratio; int checkOri;
    int dmax;                  // phase 1 keeps candidates with dist < dmax
    int wcap;                  // list capacity of one query
    int lds_ents;              // entries of a pair that phase 2 holds in LDS
    // phase-1 products (per pair)
    uint64_t* ent; int ecap; uint32_t* off; uint32_t* cnt; uint32_t* total;
    // results
    int32_t* matches12; int32_t* slot_mp; int32_t* nmatches;
};

struct WinQuery {              // wave-uniform description of one query
    bool active;
    float qx, qy, r;
    int minLevel, maxLevel;
    bool isorb;
    uint64_t d0, d1, d2, d3;
    float ur; bool has_ur;            // stereo: the query's right coordinate
};

// searched-frame record: cell (16 bits, 0xFFFF = PosInGrid false) | level (8 bits, signed) << 16 | isORB << 24
__device__ __forceinline__ uint32_t f2_info(const eorb_keypoint& k, bool isorb, const GridB& g)
{
    // Frame::PosInGrid (Frame.cc:783-793)
    const int px = (int)roundf((k.x - g.minX) * g.invW);
    const int py = (int)roundf((k.y - g.minY) * g.invH);
    const uint32_t cell = (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) ? 0xFFFFu : (uint32_t)(px * kGridRows + py);
    return cell | ((uint32_t)(kp_level(k, isorb) & 0xff) << 16) | ((uint32_t)isorb << 24);
}

// KIND 0: SearchForInitialization; 1: SearchByProjection(cur, last) / (cur, KeyFrame); 2: SearchByProjection(F, map points)
template <int KIND>
__device__ __forceinline__ WinQuery win_query(const WinArgs& A, int pair, int q)
{
    WinQuery Q;
    Q.active = true; Q.minLevel = Q.maxLevel = 0; Q.qx = Q.qy = Q.r = 0.f; Q.isorb = true; Q.d0 = Q.d1 = Q.d2 = Q.d3 = 0;
    Q.ur = 0.f; Q.has_ur = false;
    if (KIND == 0) {
        const eorb_keypoint k1 = A.kps1[(size_t)pair * A.kp1_stride + q];
        Q.isorb = A.is_orb1 ? A.is_orb1[(size_t)pair * A.capq + q] != 0 : true;
        const int level1 = kp_level(k1, Q.isorb);
        if (level1 > 0) { Q.active = false; return Q; }                                  // :730-732
        const float* PM = A.prev_matched ? A.prev_matched + ((size_t)pair * A.capq + q) * 2 : nullptr;
        Q.qx = PM ? PM[0] : k1.x; Q.qy = PM ? PM[1] : k1.y;
        Q.r = (float)A.windowSize;
        Q.minLevel = Q.maxLevel = level1;                                                // GetFeaturesInArea(x, y, windowSize, level1, level1)
        load_desc32(A.desc1 + (size_t)pair * A.desc1_slice + (size_t)q * A.dstride1, Q.d0, Q.d1, Q.d2, Q.d3);
    } else {
        if (!A.valid[q]) { Q.active = false; return Q; }
        int qlev;
        if (KIND == 2) {
            const float4 f = ((const float4*)A.qf)[q];
            Q.qx = f.x; Q.qy = f.y;
            float r = ((double)f.z > 0.998) ? 2.5f : 4.0f;            // RadiusByViewingCos (:221-227)
            if (A.th != 1.0f) r *= A.th;                                // bFactor (:49, :73-74)
            Q.r = r * f.w;
            qlev = A.qlevel[q];
            Q.minLevel = qlev - 1; Q.maxLevel = qlev;
            Q.isorb = A.mp_is_orb ? A.mp_is_orb[q] != 0 : true;
        } else {
            Q.qx = A.qf[3 * q]; Q.qy = A.qf[3 * q + 1];
            Q.isorb = A.q_is_orb ? A.q_is_orb[q] != 0 : true;
            qlev = kp_level(A.qkps[q], Q.isorb);
            Q.r = A.th * A.qf[3 * q + 2];
            if (A.mode == 1) { Q.minLevel = qlev; Q.maxLevel = -1; }
            else if (A.mode == 2) { Q.minLevel = 0; Q.maxLevel = qlev; }
            else { Q.minLevel = qlev - 1; Q.maxLevel = qlev + 1; }
        }
        const uint64_t* dq = (const uint64_t*)(A.mp_desc + (size_t)q * 32);
        Q.d0 = dq[0]; Q.d1 = dq[1]; Q.d2 = dq[2]; Q.d3 = dq[3];
        if (A.q_ur && A.uright2) { Q.ur = A.q_ur[q]; Q.has_ur = true; }
    }
    return Q;
}

// candidate test of Frame::GetFeaturesInArea (Frame.cc:747-777) + the MixedMatcher type gate (MixedMatcher.cpp:65-67, :565-568,
// :787-790).  Returns the key, or ~0 when i2 is no candidate.  key = dist << 44 | cell << 32 | index << 8 | level + 1
__device__ __forceinline__ uint64_t win_key(const WinQuery& Q, int cx0, int cx1, int cy0, int cy1, uint32_t info, float x, float y,
                                            const uint64_t* __restrict__ dp, int i2, float uright = -1.f)
{
    const int cell = (int)(info & 0xffffu);
    if (cell == 0xFFFF) return ~0ull;
    const int cx = cell / kGridRows, cy = cell - cx * kGridRows;
    if (cx < cx0 || cx > cx1 || cy < cy0 || cy > cy1) return ~0ull;
    const int lv = (int)(int8_t)((info >> 16) & 0xffu);
    if ((Q.minLevel > 0) || (Q.maxLevel >= 0)) {                     // bCheckLevels
        if (lv < Q.minLevel) return ~0ull;
        if (Q.maxLevel >= 0 && lv > Q.maxLevel) return ~0ull;
    }
    const float distx = x - Q.qx, disty = y - Q.qy;
    if (!(fabsf(distx) < Q.r && fabsf(disty) < Q.r)) return ~0ull;
    if ((((info >> 24) & 1u) != 0) != Q.isorb) return ~0ull;
    // "if(F.mvuRight[idx]>0) { er = fabs(projXR - F.mvuRight[idx]); if(er > radius) continue; }" (:96-104, :2056-2062)
    if (Q.has_ur && uright > 0.f && fabsf(Q.ur - uright) > Q.r) return ~0ull;
    const int dist = __popcll(Q.d0 ^ dp[0]) + __popcll(Q.d1 ^ dp[1]) + __popcll(Q.d2 ^ dp[2]) + __popcll(Q.d3 ^ dp[3]);
    return ((uint64_t)dist << 44) | ((uint64_t)cell << 32) | ((uint64_t)(uint32_t)i2 << 8) | (uint64_t)((lv + 1) & 0xff);
}

template <int KIND>
__global__ __launch_bounds__(1024) void win_cand_kernel(WinArgs A, int qpb)
{
    extern __shared__ unsigned char smem[];
    const int pair = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N2 = A.n2p ? A.n2p[pair] : A.n2;
    const int NQ = A.nqp ? A.nqp[pair] : A.nq;
    const int q_lo = blockIdx.x * qpb, q_hi = min(NQ, q_lo + qpb);
    if (q_lo >= NQ) return;
    const int c2 = A.cap2;
    uint64_t* d2 = (uint64_t*)smem;                                   // c2 * 4
    uint64_t* wl = d2 + (size_t)c2 * 4 + (size_t)wave * A.wcap;       // nwaves * wcap: the waves' candidate lists
    const int nwaves = (int)blockDim.x >> 6;
    float* x2 = (float*)(d2 + (size_t)c2 * 4 + (size_t)nwaves * A.wcap);
    float* y2 = x2 + c2;
    uint32_t* info2 = (uint32_t*)(y2 + c2);
    const eorb_keypoint* K2 = A.kps2 + (size_t)pair * A.kp2_stride;
    const uint8_t* D2 = A.desc2 + (size_t)pair * A.desc2_slice;
    const uint8_t* O2 = A.is_orb2 ? A.is_orb2 + (size_t)pair * A.cap2 : nullptr;
    // (the wavefront's first query is loaded with the staging loads, not behind the barrier: one global round trip less on the call's path)
#ifdef EORB_WIN_TIMING
    long long ct[6]; ct[0] = clock64();
#endif
    WinQuery Q0; Q0.active = false;
    if (q_lo + wave < q_hi) Q0 = win_query<KIND>(A, pair, q_lo + wave);
    for (int i0 = 0; i0 < N2; i0 += 2 * (int)blockDim.x) {            // two keypoints' loads in flight per thread
        eorb_keypoint k[2]; bool isorb[2]; uint64_t dd[2][4];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i = i0 + u * (int)blockDim.x + tid;
            if (i < N2) { k[u] = K2[i]; isorb[u] = O2 ? O2[i] != 0 : true; load_desc32(D2 + (size_t)i * A.dstride2, dd[u][0], dd[u][1], dd[u][2], dd[u][3]); }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i = i0 + u * (int)blockDim.x + tid;
            if (i < N2) {
                x2[i] = k[u].x; y2[i] = k[u].y; info2[i] = f2_info(k[u], isorb[u], A.g);
                d2[(size_t)i * 4 + 0] = dd[u][0]; d2[(size_t)i * 4 + 1] = dd[u][1]; d2[(size_t)i * 4 + 2] = dd[u][2]; d2[(size_t)i * 4 + 3] = dd[u][3];
            }
        }
    }
    __syncthreads();
#ifdef EORB_WIN_TIMING
    ct[1] = clock64(); ct[2] = ct[3] = ct[4] = ct[1];
#endif
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint64_t* ent = A.ent + (size_t)pair * A.ecap;
    for (int q = q_lo + wave; q < q_hi; q += nwaves) {
        const WinQuery Q = q == q_lo + wave ? Q0 : win_query<KIND>(A, pair, q);
        uint32_t n = 0;
        int cx0 = 0, cx1 = -1, cy0 = 0, cy1 = -1;
        if (Q.active && cell_range(A.g, Q.qx, Q.qy, Q.r, cx0, cx1, cy0, cy1)) {
            for (int i0 = 0; i0 < N2; i0 += 64) {
                const int i2 = i0 + lane;
                uint64_t key = ~0ull;
                if (i2 < N2) key = win_key(Q, cx0, cx1, cy0, cy1, info2[i2], x2[i2], y2[i2], &d2[(size_t)i2 * 4], i2, Q.has_ur ? A.uright2[i2] : -1.f);
                const bool ok = key != ~0ull && (int)(key >> 44) < A.dmax;
                const uint64_t bal = __ballot(ok);
                if (ok) { const uint32_t pos = n + (uint32_t)__popcll(bal & lt_mask); if (pos < (uint32_t)A.wcap) wl[pos] = key; }
                n += (uint32_t)__popcll(bal);
            }
        }
#ifdef EORB_WIN_TIMING
        ct[2] = clock64();
#endif
        uint32_t off = 0, cnt = n;
        if (n > (uint32_t)A.wcap) cnt = kWinOver;
        else if (n > 0) {
            if (lane == 0) off = atomicAdd(&A.total[pair], n) & 0x7FFFFFFFu;
            off = (uint32_t)__shfl((int)off, 0, 64);
#ifdef EORB_WIN_TIMING
            ct[3] = clock64();
#endif
            if (off + n > (uint32_t)A.ecap) cnt = kWinOver;           // the pair's pool is full
            else {
                // the list leaves sorted by key (rank = number of smaller keys; keys are unique): phase 2 then stops at the first
                // one or two candidates that pass its state filter instead of walking the whole list
                for (uint32_t j = lane; j < n; j += 64) {
                    const uint64_t kj = wl[j];
                    uint32_t rank = 0;
                    for (uint32_t i = 0; i < n; i++) rank += (wl[i] < kj) ? 1u : 0u;
                    ent[off + rank] = kj;
                }
            }
        }
        if (lane == 0) {
            A.cnt[(size_t)pair * A.capq + q] = cnt; A.off[(size_t)pair * A.capq + q] = off;
            if (cnt == kWinOver) atomicOr(&A.total[pair], 0x80000000u);          // (bit 31 of the pair's counter: some list overflowed; phase 2 reads it there)
        }
#ifdef EORB_WIN_TIMING
        ct[4] = clock64();
        if (pair == 0 && lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && (wave == 0 || wave == nwaves - 1))
            printf("win_cand<%d> wg %d wave %d (n %u): staging %lld | scan %lld | pool atomic %lld | sort + write %lld\n", KIND, (int)blockIdx.x, wave, n, ct[1] - ct[0], ct[2] - ct[1], ct[3] - ct[2], ct[4] - ct[3]);
#endif
    }
}

// wave-wide minimum of a 32-bit value in every lane: row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, row_bcast 15 / 31 across them
// (lanes without a source keep their own value), then lane 63 holds the minimum
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x)
{
#define EORB_MIN_DPP(ctrl, rmask) x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, ctrl, rmask, 0xf, false))
    EORB_MIN_DPP(0x111, 0xf); EORB_MIN_DPP(0x112, 0xf); EORB_MIN_DPP(0x114, 0xf); EORB_MIN_DPP(0x118, 0xf);
    EORB_MIN_DPP(0x142, 0xa); EORB_MIN_DPP(0x143, 0xc);
#undef EORB_MIN_DPP
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// wave-wide smallest 64-bit key (every lane gets it): minimum of the high words, then of the low words among the lanes that hold it
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t k)
{
    const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
    const uint32_t mh = wave_min_u32(hi);
    const uint32_t ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return ((uint64_t)mh << 32) | ml;
}

// wave-wide two smallest keys of the lanes' (k0 <= k1) pairs; keys are unique; result in every lane
__device__ __forceinline__ void wave_top2(uint64_t& k0, uint64_t& k1)
{
    const uint64_t b0 = wave_min_u64(k0);
    const uint64_t b1 = wave_min_u64(k0 == b0 ? k1 : k0);
    k0 = b0; k1 = b1;
}

constexpr int kWinLdsEntries = 6144;      // least number of a pair's entries that phase 2 stages in LDS (WinArgs::lds_ents: as many as fit; the rest is read from global memory)

__device__ __forceinline__ int wave_sum_i32(int v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

// histo[b] += 1 for every lane with b >= 0: one LDS atomic per distinct bin of the wavefront (matches of one frame pair share two or
// three rotation bins: lane-wise atomics on them serialise, 540 of them cost 2 us)
__device__ __forceinline__ void wave_histo_add(int* histo, int b, int lane)
{
    uint64_t todo = __ballot(b >= 0);
    while (todo) {
        const int l = __ffsll((unsigned long long)todo) - 1;
        const int bb = __shfl(b, l, 64);
        const uint64_t same = __ballot(b == bb);
        if (lane == l) atomicAdd(&histo[bb], (int)__popcll(same));
        todo &= ~same;
    }
}

// entry e of a pair: from LDS, or from the pool in global memory beyond what LDS holds.  (Written as "cond ? ents[e] : gent[e]" the
// compiler selects between the two POINTERS and emits one flat load: every walk step then goes through the flat path.)
__device__ __forceinline__ uint64_t win_entry(const uint64_t* ents, const uint64_t* __restrict__ gent, uint32_t e, uint32_t nlds)
{
    uint64_t k = ents[min(e, nlds - 1u)];
    if (e >= nlds) k = *(const volatile uint64_t*)(gent + e);        // (volatile: not to be merged with the LDS read into one flat load either)
    return k;
}

template <int KIND>
__global__ __launch_bounds__(1024) void win_resolve_kernel(WinArgs A)
{
    extern __shared__ unsigned char smem[];
    const int pair = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N2 = A.n2p ? A.n2p[pair] : A.n2;
    const int NQ = A.nqp ? A.nqp[pair] : A.nq;
    const int c2 = A.cap2, cq = A.capq;
#ifdef EORB_WIN_TIMING       // (experiment builds only: where the kernel's time goes, printed by pair 0)
    __shared__ long long s_wt[8]; __shared__ int s_wn[8];
    long long wt_last = clock64();
    if (threadIdx.x < 8) { s_wt[threadIdx.x] = 0; s_wn[threadIdx.x] = 0; }
#define WIN_T(k) do { if (threadIdx.x == 0) { const long long t_now = clock64(); s_wt[k] += t_now - wt_last; s_wn[k]++; wt_last = t_now; } } while (0)
#else
#define WIN_T(k) do { } while (0)
#endif
    // LDS carve-up
    const uint32_t nlds = (uint32_t)A.lds_ents;
    uint64_t* ents = (uint64_t*)smem;                                 // lds_ents
    uint32_t* offs = (uint32_t*)(ents + nlds);                        // cq
    uint32_t* cnts = offs + cq;                                       // cq
    int* histo = (int*)(cnts + cq);                                   // 32
    int* sh_nm = histo + 32;                                          // 4: matches, -, the fixed points' "somebody changed" words (2)
    unsigned int* sflag = (unsigned int*)(sh_nm + 2);
    int* st_a = sh_nm + 4;                                            // c2: KIND 0 vMatchedDistance, else the slot -> map point table
    int* match_at = st_a + c2;                                        // cq: candidate matched by query q when it was processed (-1: none)
    int16_t* m21 = (int16_t*)(match_at + cq);                         // c2: KIND 0 vnMatches21 (-1: none)  [cap1 < 32768]
    uint8_t* obs = (uint8_t*)(m21 + c2);                              // c2: KIND 1/2 "slot holds an observed map point"
    uint8_t* qobs = obs + c2;                                         // cq: KIND 1/2 mp_obs of the query's map point
    unsigned int* claim = (unsigned int*)(smem + (((size_t)((unsigned char*)(qobs + cq) - smem) + 3) & ~(size_t)3));   // 2 x c2 (an offset from smem, not a cast through an integer: the accesses stay LDS instructions): earliest query of the round / sweep matching the candidate
    float* ang2 = (float*)(claim + 2 * (c2 + 1));                     // c2: the searched keypoints' angles (rotation histogram)
    int* fx = (int*)(ang2 + c2);                                      // KIND 0 fixed point: 4 x cq (match / distance of the next sweep, chain links, distances)
    int32_t* M12 = A.matches12 ? A.matches12 + (size_t)pair * cq : nullptr;
    const uint64_t* gent = A.ent + (size_t)pair * A.ecap;
    const eorb_keypoint* K2s = A.kps2 + (size_t)pair * A.kp2_stride;
    // set-up: every global load that depends on nothing is issued before the first wait (a load is ~ 1 500 cycles here and the
    // loops below would take them one after the other: 11 000 cycles of 1 024 threads waiting)
    const int bd = (int)blockDim.x;
    uint32_t r_off[2], r_cnt[2]; uint8_t r_qo[2]; int r_slot[2]; float r_ang[2], r_qang[2];
    const eorb_keypoint* KQ = KIND == 0 ? A.kps1 + (size_t)pair * A.kp1_stride : A.qkps;       // the queries' keypoints (rotation histogram)
    const uint32_t total_raw = A.total[pair];
    const uint32_t total = min(total_raw & 0x7FFFFFFFu, (uint32_t)A.ecap);
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = tid + u * bd;
        r_off[u] = 0; r_cnt[u] = 0; r_qo[u] = 0; r_slot[u] = -1; r_ang[u] = 0.f; r_qang[u] = 0.f;
        if (i < NQ && KIND != 2 && A.checkOri) r_qang[u] = KQ[i].angle;
        if (i < NQ) { r_off[u] = A.off[(size_t)pair * cq + i]; r_cnt[u] = A.cnt[(size_t)pair * cq + i]; if (KIND != 0) r_qo[u] = A.mp_obs[i]; }
        if (i < N2) { if (KIND != 0) r_slot[u] = A.slot_mp[i]; if (A.checkOri) r_ang[u] = K2s[i].angle; }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = tid + u * bd;
        if (i < NQ) { offs[i] = r_off[u]; cnts[i] = r_cnt[u]; match_at[i] = -1; if (KIND != 0) qobs[i] = r_qo[u] != 0; }
    }
    for (int i = tid + 2 * bd; i < NQ; i += bd) {
        offs[i] = A.off[(size_t)pair * cq + i]; cnts[i] = A.cnt[(size_t)pair * cq + i];
        match_at[i] = -1;
        if (KIND != 0) qobs[i] = A.mp_obs[i] != 0;
    }
    {   // eight loads in flight per thread
        const uint32_t nst = min(total, nlds);
        for (uint32_t i0 = 0; i0 < nst; i0 += 8u * (uint32_t)bd) {
            uint64_t t8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t i = i0 + (uint32_t)(u * bd + tid); t8[u] = i < nst ? gent[i] : 0ull; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t i = i0 + (uint32_t)(u * bd + tid); if (i < nst) ents[i] = t8[u]; }
        }
    }
    // slot values of KIND 1 / 2: "if(F.getMapPoint(idx)) if(F.getMapPoint(idx)->Observations()>0) continue;" (:91-93, :2045-2047):
    // -1 / -3 = empty or unobserved, -2 = holds an observed map point, v >= 0 = map point v
    auto slot_in = [&](int i, int v, float ang) {
        if (KIND == 0) { st_a[i] = 0x7fffffff; m21[i] = -1; }
        else { st_a[i] = v; obs[i] = (v == -2) ? 1 : ((v >= 0) ? (A.mp_obs[v] != 0) : 0); }
        if (A.checkOri) ang2[i] = ang;
    };
#pragma unroll
    for (int u = 0; u < 2; u++) { const int i = tid + u * bd; if (i < N2) slot_in(i, r_slot[u], r_ang[u]); }
    for (int i = tid + 2 * bd; i < N2; i += bd) slot_in(i, KIND != 0 ? A.slot_mp[i] : -1, A.checkOri ? K2s[i].angle : 0.f);
    if (tid < 32) histo[tid] = 0;
    if (tid < 4) sh_nm[tid] = 0;
    const bool no_over = KIND != 2 && (total_raw >> 31) == 0u;          // (no list of the pair overflowed: phase 1 says so in the counter)
    __syncthreads();
    const bool fixpoint = KIND == 1 && no_over, fixpoint0 = KIND == 0 && no_over;
    // claims: the walk's are "earliest lane of the round" (empty = ~0); the fixed point's are tagged by their sweep (empty = 0), in two
    // buffers of c2 + 1 words: ~0 = closed for good (an observed map point in the slot; word c2 stands for "no entry")
    for (int i = tid; i < 2 * (c2 + 1); i += bd) {
        const int ii = i > c2 ? i - (c2 + 1) : i;
        claim[i] = !fixpoint || ii == c2 || (ii < N2 && obs[ii] != 0) ? 0xFFFFFFFFu : 0u;
    }
    if (KIND == 0 && !fixpoint0) for (int i = tid; i < NQ; i += bd) M12[i] = -1;      // (the walk keeps vnMatches12 in global memory)
    __syncthreads();
    WIN_T(0);

    if (fixpoint) {
        // SearchByProjection(cur, last / KeyFrame): a query takes the FIRST entry of its sorted list whose candidate is free (no
        // observed map point in the slot: :2045-2047 / :2231) if that entry is within the distance threshold, and its map point
        // then blocks the candidate for later queries when it is observed.  So query q's outcome is a function of the outcomes of
        // the queries before it, and the whole walk is the fixed point of "every query takes its first entry that no EARLIER query
        // with an observed map point has taken", iterated from "nobody has taken anything": query 0 is final after one sweep, and
        // each sweep finalises at least the first query that still changed -- in practice three to five sweeps settle a window of
        // 1 024 queries (dependency chains are short), where the walk below spends a round per conflict.
        // One barrier per sweep: the claims of sweep s go to buffer s & 1 tagged with s (atomicMax of s << 16 | 0xFFFF - tid: a newer
        // sweep beats what the buffer still holds, the earliest query wins inside a sweep), sweep s reads buffer (s - 1) & 1 and takes
        // only tags s - 1, nobody clears anything; "somebody changed" is the word sflag[s & 1] == s.
        int nm = 0;
        uint32_t s = 1;
        for (int base = 0; base < NQ; base += (int)blockDim.x) {
            const int q = base + tid;
            const uint32_t c = q < NQ ? cnts[q] : 0u;
            const uint32_t o = q < NQ ? offs[q] : 0u;
            const bool blocks = q < NQ && qobs[q] != 0;
            int pick = -1;
            // A sweep lasts as long as its slowest wavefront, and sixteen of them share the four SIMDs of one CU: what counts is the
            // instructions of a sweep.  Per entry one LDS word and one compare: a candidate is free for query tid in sweep s iff its
            // word is <= T = (s - 1) << 16 | 0xFFFF - tid (older tags are smaller, a claimant after tid has a smaller low half, ~0 is
            // a closed slot); the tests on the distance are vacuous here (phase 1 listed dist < dmax <= min(dist_th + 1, 256)).  The
            // candidates of the first eight entries (the longest walks of a crowded frame are about that long) stay in registers.
            auto cand4 = [&](uint32_t j0, uint32_t* a4) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    a4[u] = (uint32_t)c2;
                    if (j0 + (uint32_t)u < c) a4[u] = (uint32_t)((win_entry(ents, gent, o + j0 + (uint32_t)u, nlds) >> 8) & 0xffffffu);
                }
            };
            uint32_t f4[4], g4[4];
            cand4(0, f4); cand4(4, g4);
            for (;; s++) {
                unsigned int* wr = claim + ((s & 1u) ? (uint32_t)(c2 + 1) : 0u);            // buffer s & 1
                const unsigned int* rd = claim + ((s & 1u) ? 0u : (uint32_t)(c2 + 1));      // buffer (s - 1) & 1
                const unsigned int T = ((s - 1u) << 16) | (0xFFFFu - (unsigned int)tid);
                int np = -1;
                {
                    const unsigned int v0 = rd[f4[0]], v1 = rd[f4[1]], v2 = rd[f4[2]], v3 = rd[f4[3]];
                    np = v0 <= T ? (int)f4[0] : v1 <= T ? (int)f4[1] : v2 <= T ? (int)f4[2] : v3 <= T ? (int)f4[3] : -1;
                }
                if (np < 0 && c > 4u) {
                    const unsigned int v0 = rd[g4[0]], v1 = rd[g4[1]], v2 = rd[g4[2]], v3 = rd[g4[3]];
                    np = v0 <= T ? (int)g4[0] : v1 <= T ? (int)g4[1] : v2 <= T ? (int)g4[2] : v3 <= T ? (int)g4[3] : -1;
                }
                for (uint32_t j0 = 8; np < 0 && j0 < c; j0 += 4) {
                    uint32_t a4[4];
                    cand4(j0, a4);
                    const unsigned int v0 = rd[a4[0]], v1 = rd[a4[1]], v2 = rd[a4[2]], v3 = rd[a4[3]];
                    np = v0 <= T ? (int)a4[0] : v1 <= T ? (int)a4[1] : v2 <= T ? (int)a4[2] : v3 <= T ? (int)a4[3] : -1;
                }
                if (np >= 0 && blocks) atomicMax(&wr[np], (s << 16) | (0xFFFFu - (unsigned int)tid));
                if (np != pick) sflag[s & 1u] = s;
                pick = np;
                __syncthreads();
                WIN_T(1);
                if (sflag[s & 1u] != s) break;
            }
            s += 2;                                                     // (the next window must not meet this one's last claims)
            // the window's commits: the LAST query that took a candidate owns its slot; an observed map point closes it
            if (pick >= 0) { match_at[q] = pick; st_a[pick] = -1; nm++; }      // (the slot's old content -- empty or an unobserved map point -- goes)
            __syncthreads();
            if (pick >= 0) atomicMax(&st_a[pick], q);
            if (pick >= 0 && blocks) { claim[pick] = 0xFFFFFFFFu; claim[c2 + 1 + pick] = 0xFFFFFFFFu; }
            __syncthreads();
            WIN_T(2);
        }
        nm = wave_sum_i32(nm);
        if (lane == 0 && nm) atomicAdd(&sh_nm[0], nm);
    } else if (fixpoint0) {
        // SearchForInitialization (:714-831): query q skips a candidate whose vMatchedDistance is <= its distance (:755), i.e. a candidate
        // that an EARLIER query matched at a distance <= its own; among the rest its best / second best decide whether it matches its
        // best (:770-772), which then carries q's distance and loses its previous owner (:774-781).  So q's outcome is a function of the
        // matches of the queries before it: D_q[i2] = min { dist(q', i2) : q' < q matched i2 }, and the walk is the fixed point of
        // "every query decides against the D of the previous sweep's matches", iterated from "nobody matched": query 0 is final after
        // one sweep, every sweep finalises at least the first query that still changed; dependency chains are short in practice (a
        // handful of sweeps for 1 000 queries, where the round-based walk below took a round per conflict: 49 us).  Per candidate the
        // sweep's matchers hang on a chain (head in st_a, links in fx): the minimum over the earlier ones is a walk of one or two links.
        int* mcur = match_at; int* mnew = fx; int* dcur = fx + cq; int* dnew = fx + 2 * cq; int* nxt = fx + 3 * cq;
        for (int i = tid; i < N2; i += blockDim.x) st_a[i] = -1;              // chain heads
        for (int i = tid; i < NQ; i += blockDim.x) dcur[i] = 0;
        __syncthreads();
        for (int sweep = 0; sweep <= NQ; sweep++) {
            int changed = 0;
            for (int q = tid; q < NQ; q += blockDim.x) {
                const uint32_t c = cnts[q], o = offs[q];
                uint64_t k0 = ~0ull, k1 = ~0ull;
                int found = 0;
                for (uint32_t j = 0; j < c && found < 2; j++) {
                    const uint32_t e = o + j;
                    const uint64_t key = win_entry(ents, gent, e, nlds);
                    const int idx = (int)((key >> 8) & 0xffffffu), dist = (int)(key >> 44);
                    int D = 0x7fffffff;                                       // vMatchedDistance[idx] as query q finds it
                    for (int p2 = st_a[idx]; p2 >= 0; p2 = nxt[p2]) if (p2 < q) D = min(D, dcur[p2]);
                    if (D > dist) { if (found == 0) k0 = key; else k1 = key; found++; }
                }
                int m = -1, d = 0;
                if (k0 != ~0ull) {
                    const int bestDist = (int)(k0 >> 44), bestDist2 = (k1 == ~0ull) ? 0x7fffffff : (int)(k1 >> 44);
                    if (bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * A.nnratio) { m = (int)((k0 >> 8) & 0xffffffu); d = bestDist; }   // :770-772
                }
                mnew[q] = m; dnew[q] = d;
                changed |= (m != mcur[q]) ? 1 : 0;
            }
            if (changed) sflag[sweep & 1] = (unsigned int)sweep + 1u;         // ("somebody changed" as a tagged word: one plain barrier)
            __syncthreads();
            WIN_T(1);
            if (sflag[sweep & 1] != (unsigned int)sweep + 1u) break;          // (mnew == mcur: the chains already describe the final matches)
            { int* t = mcur; mcur = mnew; mnew = t; t = dcur; dcur = dnew; dnew = t; }
            for (int i = tid; i < N2; i += blockDim.x) st_a[i] = -1;
            __syncthreads();
            for (int q = tid; q < NQ; q += blockDim.x) { const int m = mcur[q]; if (m >= 0) nxt[q] = atomicExch(&st_a[m], q); }
            __syncthreads();
        }
        // the walk's final state: every matched query pushed its rotation-histogram entry (match_at), a candidate belongs to the LAST of
        // its matchers (each later one came with a strictly smaller distance and took it over, :774-781)
        if (mcur != match_at) { for (int q = tid; q < NQ; q += blockDim.x) match_at[q] = mcur[q]; }
        __syncthreads();
        for (int q = tid; q < NQ; q += blockDim.x) fx[q] = -1;                // vnMatches12 while the kernel still works on it
        __syncthreads();
        int nm = 0;
        for (int i = tid; i < N2; i += blockDim.x) {
            int owner = -1;
            for (int p2 = st_a[i]; p2 >= 0; p2 = nxt[p2]) owner = max(owner, p2);
            if (owner >= 0) { fx[owner] = i; nm++; }
        }
        nm = wave_sum_i32(nm);
        if (lane == 0 && nm) atomicAdd(&sh_nm[0], nm);
    } else if (wave == 0) {
        // The queries are walked in order, 64 at a time (lane = query).  Every lane evaluates its query against the committed
        // state; a query commits in this round only if no EARLIER query of the round changes a candidate it depends on.  The state
        // only ever removes candidates (vMatchedDistance decreases, a slot turns "observed"), and a query's outcome depends on its
        // best and second-best passing candidates alone: so query j must wait exactly when an earlier query of the round matches
        // j's best or second-best candidate.  The round commits the conflict-free prefix (never empty: its first query has no
        // predecessor) and the next round starts at the first conflicting query.  Committed queries of one round have distinct best
        // candidates, so their updates are independent.
        int nm = 0;
        const eorb_keypoint* K2 = A.kps2 + (size_t)pair * A.kp2_stride;
        const uint8_t* D2 = A.desc2 + (size_t)pair * A.desc2_slice;
        const uint8_t* O2 = A.is_orb2 ? A.is_orb2 + (size_t)pair * A.cap2 : nullptr;
        auto passes = [&](uint64_t key) -> bool {
            const int idx = (int)((key >> 8) & 0xffffffu);
            const int dist = (int)(key >> 44);
            return (KIND == 0) ? (st_a[idx] > dist)                     // "if(vMatchedDistance[i2]<=dist) continue;" :755
                               : (obs[idx] == 0 && dist < 256);         // bestDist starts at 256, strict '<'
        };
        // outcome of a query from its two smallest passing keys: does it match its best candidate?
        auto matched = [&](uint64_t k0, uint64_t k1) -> bool {
            if (k0 == ~0ull) return false;
            const int bestDist = (int)(k0 >> 44), bestLevel = (int)(k0 & 0xff) - 1;
            if (KIND == 0) {
                const int bestDist2 = (k1 == ~0ull) ? 0x7fffffff : (int)(k1 >> 44);
                return bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * A.nnratio;       // :770-772
            } else if (KIND == 2) {
                int bestDist2 = 256, bestLevel2 = -1;
                if (k1 != ~0ull) { bestDist2 = (int)(k1 >> 44); bestLevel2 = (int)(k1 & 0xff) - 1; }
                if (bestDist > TH_HIGH) return false;                                              // :131-147
                const bool reject = (bestLevel == bestLevel2) && ((float)bestDist > A.nnratio * (float)bestDist2);
                return !reject && (bestLevel != bestLevel2 || (float)bestDist <= A.nnratio * (float)bestDist2);
            } else {
                return bestDist <= A.dist_th;                                                      // :2140 / :2271
            }
        };
        // the greedy update of one matched query (one lane; distinct candidates across the lanes of a round)
        auto commit = [&](int q, uint64_t k0, int& stolen) {
            const int bestIdx = (int)((k0 >> 8) & 0xffffffu);
            if (KIND == 0) {
                const int old = m21[bestIdx];
                if (old >= 0) { M12[old] = -1; stolen = 1; }                                       // :774-778
                M12[q] = bestIdx; m21[bestIdx] = (int16_t)q; st_a[bestIdx] = (int)(k0 >> 44); match_at[q] = bestIdx;
            } else {
                st_a[bestIdx] = q; obs[bestIdx] = qobs[q];
                if (KIND == 1) match_at[q] = bestIdx;
            }
        };
        int pos = 0;
        while (pos < NQ) {
            const int q = pos + lane;
            const uint32_t c = (q < NQ) ? cnts[q] : 0u;
            const uint64_t over = __ballot(c == kWinOver);
            const int limit = over ? (__ffsll((unsigned long long)over) - 1) : 64;
            if (limit == 0) {
                // the list of query `pos` overflowed: the wavefront scans the whole searched frame (global memory) for it
                uint64_t k0 = ~0ull, k1 = ~0ull;
                const WinQuery Q = win_query<KIND>(A, pair, pos);
                int cx0, cx1, cy0, cy1;
                if (Q.active && cell_range(A.g, Q.qx, Q.qy, Q.r, cx0, cx1, cy0, cy1)) {
                    for (int i2 = lane; i2 < N2; i2 += 64) {
                        const eorb_keypoint k = K2[i2];
                        const bool isorb = O2 ? O2[i2] != 0 : true;
                        uint64_t dd[4];
                        load_desc32(D2 + (size_t)i2 * A.dstride2, dd[0], dd[1], dd[2], dd[3]);
                        const uint64_t key = win_key(Q, cx0, cx1, cy0, cy1, f2_info(k, isorb, A.g), k.x, k.y, dd, i2, Q.has_ur ? A.uright2[i2] : -1.f);
                        if (key != ~0ull && passes(key)) { if (key < k0) { k1 = k0; k0 = key; } else if (key < k1) k1 = key; }
                    }
                }
                wave_top2(k0, k1);
                if (matched(k0, k1)) {
                    int stolen = 0;
                    if (lane == 0) commit(pos, k0, stolen);
                    stolen = __shfl(stolen, 0, 64);
                    nm += 1 - stolen;
                }
                pos += 1;
                continue;
            }
            // ---- evaluate (lane = query) ----
            uint64_t k0 = ~0ull, k1 = ~0ull;
            const bool live = lane < limit && c != 0u;
            if (live) {
                const uint32_t o = offs[q];
                // the list is sorted: the first passing entries are the best and the second best
                constexpr int need = (KIND == 1) ? 1 : 2;
                int found = 0;
                // the first four entries and their states in flight together (a round is a chain of LDS round trips: the two best
                // passing entries are almost always among them), the rest of the list one by one
                uint64_t e4[4]; bool p4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t e = o + (uint32_t)u; e4[u] = (uint32_t)u < c ? win_entry(ents, gent, e, nlds) : ~0ull; }
#pragma unroll
                for (int u = 0; u < 4; u++) p4[u] = e4[u] != ~0ull && passes(e4[u]);
#pragma unroll
                for (int u = 0; u < 4; u++) if (p4[u] && found < need) { if (found == 0) k0 = e4[u]; else k1 = e4[u]; found++; }
                for (uint32_t j = 4; j < c && found < need; j++) {
                    const uint32_t e = o + j;
                    const uint64_t key = win_entry(ents, gent, e, nlds);
                    if (passes(key)) { if (found == 0) k0 = key; else k1 = key; found++; }
                }
            }
            const bool mt = live && matched(k0, k1);
            const int i0 = (int)((k0 >> 8) & 0xffffffu), i1 = (int)((k1 >> 8) & 0xffffffu);
            // ---- claims: the earliest query of the round that changes each candidate ----
            if (mt) atomicMin(&claim[i0], (unsigned int)lane);
            bool conflict = false;
            if (live) conflict = (k0 != ~0ull && claim[i0] < (unsigned int)lane) || (k1 != ~0ull && claim[i1] < (unsigned int)lane);
            const uint64_t cm = __ballot(conflict);
            const int ncommit = cm ? (__ffsll((unsigned long long)cm) - 1) : limit;
            int stolen = 0;
            const bool doit = mt && lane < ncommit;
            if (doit) commit(q, k0, stolen);
            if (mt) claim[i0] = 0xFFFFFFFFu;
            nm += (int)__popcll(__ballot(doit)) - (int)__popcll(__ballot(stolen != 0));
            pos += ncommit;
        }
        if (lane == 0) sh_nm[0] = nm;
    }
    __syncthreads();
    WIN_T(3);
    if (KIND == 0) {
        int8_t* bin1 = (int8_t*)qobs;                                  // free in this kind: the bin of query q's push
        int* m12s = fx;                                                // vnMatches12 in LDS until the last loop
        if (!fixpoint0) { for (int i = tid; i < NQ; i += bd) m12s[i] = M12[i]; __syncthreads(); }
        if (A.checkOri) {
            // rotHist[bin].push_back(i1) happened for every match when it was made (:785-797), stolen ones included
            auto push = [&](int i, float qang, bool have) {                // (called by whole wavefronts)
                int b = -1;
                if (i < NQ) { const int m = match_at[i]; if (m >= 0) b = rot_bin(have ? qang : KQ[i].angle, ang2[m]); bin1[i] = (int8_t)b; }
                wave_histo_add(histo, b, lane);
            };
            push(tid, r_qang[0], true);
            if (bd < NQ) push(tid + bd, r_qang[1], true);
            for (int i0 = 2 * bd; i0 < NQ; i0 += bd) push(i0 + tid, 0.f, false);
            __syncthreads();
            int ind1, ind2, ind3;
            three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
            int dec = 0;
            for (int i = tid; i < NQ; i += bd) {
                const int b = bin1[i];
                if (b >= 0 && b != ind1 && b != ind2 && b != ind3 && m12s[i] >= 0) { m12s[i] = -1; dec++; }
            }
            dec = wave_sum_i32(dec);
            if (lane == 0 && dec) atomicSub(&sh_nm[0], dec);
            __syncthreads();
        }
        float* PM = A.prev_matched ? A.prev_matched + (size_t)pair * cq * 2 : nullptr;
        for (int i = tid; i < NQ; i += bd) {
            const int m = m12s[i];
            M12[i] = m;
            if (PM && m >= 0) { const eorb_keypoint k = K2s[m]; PM[2 * i] = k.x; PM[2 * i + 1] = k.y; }
        }
    } else {
        if (KIND == 1 && A.checkOri) {
            // rotHist[bin].push_back(bestIdx2) per match (:2146-2160); losing bins: setMapPoint(idx, NULL), nmatches-- per push
            unsigned int* hbin = (unsigned int*)ents;                  // the entry buffer is free now: c2 words (c2 <= 2 * kWinLdsEntries <= 2 * lds_ents)
            for (int i = tid; i < N2; i += bd) hbin[i] = 0u;
            __syncthreads();
            auto push = [&](int i, float qang, bool have) {                // (called by whole wavefronts)
                int b = -1;
                if (i < NQ) {
                    const int m = match_at[i];
                    if (m >= 0) { b = rot_bin(have ? qang : KQ[i].angle, ang2[m]); atomicOr(&hbin[m], 1u << b); }
                }
                wave_histo_add(histo, b, lane);
            };
            push(tid, r_qang[0], true);
            if (bd < NQ) push(tid + bd, r_qang[1], true);
            for (int i0 = 2 * bd; i0 < NQ; i0 += bd) push(i0 + tid, 0.f, false);
            __syncthreads();
            int ind1, ind2, ind3;
            three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
            unsigned int keep = 0u;
            if (ind1 >= 0) keep |= 1u << ind1;
            if (ind2 >= 0) keep |= 1u << ind2;
            if (ind3 >= 0) keep |= 1u << ind3;
            for (int i = tid; i < N2; i += bd)
                if (hbin[i] & ~keep) st_a[i] = -1;
            if (wave == 0) {
                const int dec = wave_sum_i32((lane < HISTO_LENGTH && !(keep & (1u << lane))) ? histo[lane] : 0);
                if (lane == 0) sh_nm[0] -= dec;
            }
            __syncthreads();
        }
        for (int i = tid; i < N2; i += bd) A.slot_mp[i] = st_a[i];
    }
    if (tid == 0) { A.nmatches[pair] = sh_nm[0]; A.total[pair] = 0u; }        // (the counter goes back to zero for the next call's phase 1)
#ifdef EORB_WIN_TIMING
    WIN_T(4);
    if (tid == 0 && pair == 0)
        printf("win_resolve<%d> NQ %d N2 %d entries %u: set-up %lld | sweeps %d x %lld | commits %d x %lld | tail of the walk %lld | histogram + outputs %lld (clock64 ticks)\n", KIND, NQ, N2, total,
               s_wt[0], s_wn[1], s_wn[1] ? s_wt[1] / s_wn[1] : 0, s_wn[2], s_wn[2] ? s_wt[2] / s_wn[2] : 0, s_wt[3], s_wt[4]);
#endif
}

static size_t win_cand_lds(int cap2, int wcap, int nwaves) { return ((size_t)cap2 * (32 + 4 + 4 + 4) + (size_t)nwaves * wcap * 8 + 15) & ~(size_t)15; }
// phase 2's LDS without the entries: offs, cnts, match_at, qobs, fx per query; st_a, m21, obs, two claim buffers, angle per searched keypoint
static size_t win_resolve_lds_rest(int cap2, int capq)
{
    return ((size_t)capq * (4 + 4 + 4 + 1 + 16) + (32 + 4) * 4 + (size_t)cap2 * (4 + 2 + 1 + 8 + 4) + 16 + 15) & ~(size_t)15;
}

// smallest d in [lo, 256] for which pred(d) holds, else 257 (= keep every candidate)
template <typename F> static int first_dist(int lo, F pred) { for (int d = lo; d <= 256; d++) if (pred(d)) return d; return 257; }

template <int KIND>
static int launch_win(eorb_ctx* c, WinArgs& A, int npairs, int nq_max, const char* name)
{
    if (A.cap2 >= (1 << 24) || A.capq >= 32768) return set_err(c, EORB_E_CAPACITY, "%s: too many keypoints", name);
    if (A.cap2 > 2 * kWinLdsEntries) return set_err(c, EORB_E_CAPACITY, "%s: %d keypoints in the searched frame (limit %d)", name, A.cap2, 2 * kWinLdsEntries);
    A.wcap = c->dbg_win_wcap > 0 ? c->dbg_win_wcap : 512;
    A.ecap = c->dbg_win_ecap > 0 ? c->dbg_win_ecap : std::max(4096, 16 * A.capq);
    // phase 2 keeps as many of a pair's entries in LDS as fit beside its tables (one read from global memory inside a sweep is what the
    // whole workgroup then waits for at the sweep's barrier), at least kWinLdsEntries
    // phase 1: every workgroup stages the searched frame (44 B per keypoint) from the L2 before its wavefronts take a query each.  A lone
    // pair (one frame per call) is latency: 16 wavefronts per workgroup stage it in two loads per thread and 4 x fewer workgroups do so
    // (SearchByProjection 24 -> ... us); a batch of pairs fills the chip either way and keeps the small workgroups.
    static const int cand_env = [] { const char* e = getenv("EORB_WIN_CAND_WAVES"); return e ? atoi(e) : 0; }();      // (A/B runs)
    int cw = cand_env > 0 ? cand_env : (npairs >= 8 ? 4 : 16);
    while (cw > 4 && win_cand_lds(A.cap2, A.wcap, cw) > 159 * 1024) cw >>= 1;
    const size_t lds1 = win_cand_lds(A.cap2, A.wcap, cw), rest2 = win_resolve_lds_rest(A.cap2, A.capq);
    A.lds_ents = (int)std::min<size_t>((size_t)A.ecap, rest2 < 159 * 1024 ? (159 * 1024 - rest2) / 8 : 0);
    const bool fits = A.lds_ents >= std::min(A.ecap, kWinLdsEntries);
    if (c->dbg_win_lds_ents > 0) A.lds_ents = std::min(A.lds_ents, std::max(c->dbg_win_lds_ents, (A.cap2 + 1) / 2));      // (the rotation histogram's bin masks reuse the buffer)
    const size_t lds2 = (size_t)A.lds_ents * 8 + rest2;
    if (lds1 > 159 * 1024 || !fits)        // (the kernels own a few words of static LDS besides)
        return set_err(c, EORB_E_CAPACITY, "%s: %zu / %zu B of LDS needed (searched frame %d, queries %d)", name, lds1, lds2, A.cap2, A.capq);
    int rc;
    // phase-1 products: entries | off | cnt | total
    const size_t ent_b = sizeof(uint64_t) * (size_t)npairs * A.ecap, oc_b = sizeof(uint32_t) * (size_t)npairs * A.capq;
    if ((rc = ensure(c, c->win_ws, ent_b + 2 * oc_b + 64))) return rc;
    A.ent = (uint64_t*)c->win_ws.p;
    A.off = (uint32_t*)((char*)c->win_ws.p + ent_b);
    A.cnt = A.off + (size_t)npairs * A.capq;
    // the pairs' entry counters start at zero: phase 2 leaves them so (a fill per call is a dependent launch of its own: 4 + 3 us)
    if (c->win_total_n < (size_t)npairs) {
        if ((rc = ensure(c, c->win_total, sizeof(uint32_t) * (size_t)npairs))) return rc;
        EORB_HIP(c, hipMemsetAsync(c->win_total.p, 0, c->win_total.cap, c->stream));
        c->win_total_n = c->win_total.cap / sizeof(uint32_t);
    }
    A.total = (uint32_t*)c->win_total.p;
    const size_t total_n = c->win_total_n;
    c->win_total_n = 0;                                 // (until both phases are known to have been launched)
    // per device, not per process: a second context on another GPU needs the opt-in too -- once per context and kind
    if (!(c->win_attr_done & (1u << KIND))) {
        EORB_HIP(c, hipFuncSetAttribute((const void*)win_cand_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        EORB_HIP(c, hipFuncSetAttribute((const void*)win_resolve_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        c->win_attr_done |= 1u << KIND;
    }
    ProfScope ps(c, name);
    // queries per phase-1 workgroup: a lone pair is spread over as many workgroups as it has queries per wavefront (one each: 0.091 ->
    // 0.085 ms per SearchForInitialization call against two each; every workgroup stages the searched frame, 57 KB, from the L2)
    static const int qpb_env = [] { const char* e = getenv("EORB_WIN_QPB"); return e ? atoi(e) : 0; }();      // (A/B runs)
    const int qpb = qpb_env > 0 ? qpb_env : (npairs >= 8 ? 32 : cw);
    win_cand_kernel<KIND><<<dim3((nq_max + qpb - 1) / qpb, npairs), 64 * cw, lds1, c->stream>>>(A, qpb);
    // (the fixed points of SearchForInitialization and SearchByProjection(cur, last) settle a window of blockDim queries at a time: the widest block)
    win_resolve_kernel<KIND><<<npairs, KIND == 2 ? 256 : 1024, lds2, c->stream>>>(A);
    EORB_LAUNCH_CHECK(c, name);
    c->win_total_n = total_n;
    return EORB_OK;
}

int search_init_dev(eorb_ctx* c, int npairs,
                    const eorb_keypoint* kps1, const int32_t* n1, size_t kp1_stride, const uint8_t* desc1, int dstride1, size_t desc1_slice,
                    const uint8_t* is_orb1,
                    const eorb_keypoint* kps2, const int32_t* n2, size_t kp2_stride, const uint8_t* desc2, int dstride2, size_t desc2_slice,
                    const uint8_t* is_orb2, int cap1, int cap2,
                    eorb_grid_bounds gb, float* prev_matched, int32_t* matches12, int windowSize, float nnratio,
                    int checkOri, int32_t* nmatches)
{
    if (npairs <= 0) return EORB_OK;
    WinArgs A{};
    A.kps2 = kps2; A.kp2_stride = kp2_stride; A.n2p = n2; A.desc2 = desc2; A.dstride2 = dstride2; A.desc2_slice = desc2_slice;
    A.is_orb2 = is_orb2; A.cap2 = cap2;
    A.nqp = n1; A.capq = cap1;
    A.kps1 = kps1; A.kp1_stride = kp1_stride; A.desc1 = desc1; A.dstride1 = dstride1; A.desc1_slice = desc1_slice; A.is_orb1 = is_orb1;
    A.prev_matched = prev_matched; A.windowSize = windowSize;
    A.g = GridB{gb.minX, gb.minY, gb.invW, gb.invH}; A.nnratio = nnratio; A.checkOri = checkOri;
    A.matches12 = matches12; A.nmatches = nmatches;
    // a second-best candidate only matters while "bestDist < bestDist2 * ratio" (:772) can fail for some bestDist <= TH_LOW: every
    // dist2 with TH_LOW < dist2 * ratio passes like INT_MAX does (float product monotone in dist2); best candidates need dist <= TH_LOW
    A.dmax = first_dist(TH_LOW + 1, [&](int d) { return (float)TH_LOW < (float)d * nnratio; });
    return launch_win<0>(c, A, npairs, cap1, "search_init");
}

int search_proj_last_dev(eorb_ctx* c, const eorb_keypoint* cur_kps, int n_cur, const uint8_t* cur_desc, int cur_stride,
                         const uint8_t* cur_is_orb, const eorb_keypoint* last_kps, int n_last, const uint8_t* last_is_orb,
                         const uint8_t* valid, const float* uvs, const uint8_t* mp_desc, const uint8_t* mp_obs,
                         int dist_th, eorb_grid_bounds gb, int32_t* cur_mp, float th, int mode, int checkOri,
                         int32_t* nmatches, const float* cur_uright, const float* q_ur)
{
    WinArgs A{};
    A.dist_th = dist_th; A.uright2 = cur_uright; A.q_ur = q_ur;
    A.kps2 = cur_kps; A.n2 = n_cur; A.cap2 = std::max(n_cur, 1); A.desc2 = cur_desc; A.dstride2 = cur_stride; A.is_orb2 = cur_is_orb;
    A.nq = n_last; A.capq = std::max(n_last, 1); A.qkps = last_kps; A.q_is_orb = last_is_orb; A.valid = valid; A.qf = uvs;
    A.mp_desc = mp_desc; A.mp_obs = mp_obs;
    A.g = GridB{gb.minX, gb.minY, gb.invW, gb.invH};
    A.slot_mp = cur_mp; A.th = th; A.mode = mode; A.checkOri = checkOri; A.nmatches = nmatches;
    A.dmax = dist_th < 0 ? 0 : std::min(dist_th + 1, 256);          // best only: "if(bestDist<=TH_HIGH)" (:2140), bestDist starts at 256: every listed candidate
                                                                    // passes both tests, which phase 2's fixed point relies on
    return launch_win<1>(c, A, 1, n_last, "search_proj_last");
}

int search_proj_map_dev(eorb_ctx* c, const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const uint8_t* is_orb,
                        int M, const uint8_t* in_view, const float4* mp_f4, const int32_t* level, const uint8_t* mp_desc,
                        const uint8_t* mp_obs, const uint8_t* mp_is_orb, eorb_grid_bounds gb, int32_t* frame_mp, float th,
                        float nnratio, int32_t* nmatches, const float* uright, const float* q_ur)
{
    WinArgs A{};
    A.uright2 = uright; A.q_ur = q_ur;
    A.kps2 = kps; A.n2 = n; A.cap2 = std::max(n, 1); A.desc2 = desc; A.dstride2 = stride; A.is_orb2 = is_orb;
    A.nq = M; A.capq = std::max(M, 1); A.valid = in_view; A.qf = (const float*)mp_f4; A.qlevel = level;
    A.mp_desc = mp_desc; A.mp_obs = mp_obs; A.mp_is_orb = mp_is_orb;
    A.g = GridB{gb.minX, gb.minY, gb.invW, gb.invH};
    A.slot_mp = frame_mp; A.th = th; A.nnratio = nnratio; A.nmatches = nmatches;
    // the second best rejects only while "bestDist > ratio * bestDist2" (:134) can hold for some bestDist <= TH_HIGH
    A.dmax = std::min(256, first_dist(TH_HIGH + 1, [&](int d) { return !((float)TH_HIGH > nnratio * (float)d); }));
    return launch_win<2>(c, A, 1, M, "search_proj_map");
}


// ---------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (:276-478), mono branch.  A frame feature belongs to
// exactly one vocabulary node, so the greedy state (vpMapPointMatches) never crosses nodes: shared nodes are matched
// concurrently (one wavefront each), KeyFrame features of a node sequentially, the node's frame features over the lanes.
struct BowArgs {
    const eorb_keypoint* kf_kps; const uint8_t* kf_desc; const uint8_t* kf_has_mp;
    const uint32_t* kf_nodes; const int32_t* kf_off; const int32_t* kf_idx; int kf_nn;
    const eorb_keypoint* f_kps; int n_f; const uint8_t* f_desc;
    const uint32_t* f_nodes; const int32_t* f_off; const int32_t* f_idx; int f_nn;
    int32_t* match_f; int8_t* bin_f; int32_t* histo; int32_t* nmatches;
    float nnratio; int checkOri;
    int kf_kf;                       // 1: SearchByBoW(KF, KF) (:833-973): output per idx1, vbMatched2 flags, strict TH_LOW
    const uint8_t* f_has_mp; int32_t* match12; int n_kf;
};

__global__ __launch_bounds__(256) void search_bow_kernel(BowArgs A)
{
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (int a = gw; a < A.kf_nn; a += nw) {
        // the frame node with the same id (both lists ascending): binary search = the reference's lower_bound walk
        const uint32_t node = A.kf_nodes[a];
        // (node ids are unique in a FeatureVector: the lanes look at 64 positions at a time -- one memory round trip, not the seven
        // dependent ones of a binary search)
        int lo = -1;
        for (int base = 0; base < A.f_nn && lo < 0; base += 64) {
            const int pos = base + lane;
            const uint64_t hit = __ballot(pos < A.f_nn && A.f_nodes[pos] == node);
            if (hit) lo = base + __ffsll((unsigned long long)hit) - 1;
        }
        if (lo < 0) continue;
        const int f0 = A.f_off[lo], f1 = A.f_off[lo + 1];
        const int k0n = A.kf_off[a], k1n = A.kf_off[a + 1];
        if (f1 - f0 <= 64 && k1n - k0n <= 64) {
            // The usual node (about ten features on either side): everything it touches is loaded once -- lane = frame feature for the
            // candidates, lane = KeyFrame feature for the queries, handed round by shuffles.  A frame feature belongs to this node
            // only, so "already matched" (vpMapPointMatches[realIdxF] / vbMatched2) is a lane-local flag: no memory round trip between
            // the KeyFrame features of the node (the walk below took three dependent global loads per feature).
            const int nF = f1 - f0, nK = k1n - k0n;
            int idxF = -1; bool availF = false; uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            if (lane < nF) {
                idxF = A.f_idx[f0 + lane];
                availF = A.match_f[idxF] < 0 && !(A.kf_kf && !A.f_has_mp[idxF]);
                const uint64_t* tp = (const uint64_t*)(A.f_desc + (size_t)idxF * 32);
                t0 = tp[0]; t1 = tp[1]; t2 = tp[2]; t3 = tp[3];
            }
            int idxK = -1; bool hasK = false; uint64_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
            if (lane < nK) {
                idxK = A.kf_idx[k0n + lane];
                hasK = A.kf_has_mp[idxK] != 0;
                const uint64_t* dq = (const uint64_t*)(A.kf_desc + (size_t)idxK * 32);
                q0 = dq[0]; q1 = dq[1]; q2 = dq[2]; q3 = dq[3];
            }
            for (int i = 0; i < nK; i++) {
                if (!__shfl((int)hasK, i, 64)) continue;
                const uint64_t b0 = __shfl(q0, i, 64), b1 = __shfl(q1, i, 64), b2 = __shfl(q2, i, 64), b3 = __shfl(q3, i, 64);
                const int realIdxKF = __shfl(idxK, i, 64);
                uint64_t k0 = ~0ull, k1 = ~0ull;
                if (availF) {
                    const int dist = __popcll(b0 ^ t0) + __popcll(b1 ^ t1) + __popcll(b2 ^ t2) + __popcll(b3 ^ t3);
                    k0 = ((uint64_t)dist << 32) | (uint32_t)lane;          // candidate order = vector order = lane order
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const uint64_t o0 = __shfl_xor(k0, d, 64), o1 = __shfl_xor(k1, d, 64);
                    const uint64_t l0 = k0 < o0 ? k0 : o0, h0 = k0 < o0 ? o0 : k0, s1 = k1 < o1 ? k1 : o1;
                    k0 = l0; k1 = h0 < s1 ? h0 : s1;
                }
                if (k0 != ~0ull && (int)(k0 >> 32) < 256) {
                    const int bestDist1 = (int)(k0 >> 32);
                    const int bestDist2 = (k1 != ~0ull && (int)(k1 >> 32) < 256) ? (int)(k1 >> 32) : 256;
                    if ((A.kf_kf ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW) && (float)bestDist1 < A.nnratio * (float)bestDist2) {
                        const int win = (int)(k0 & 63u);
                        const int bestIdxF = __shfl(idxF, win, 64);
                        if (lane == win) availF = false;
                        if (lane == 0) {
                            A.match_f[bestIdxF] = realIdxKF;
                            atomicAdd(A.nmatches, 1);
                            if (A.kf_kf) A.match12[realIdxKF] = bestIdxF;
                            if (A.checkOri) {
                                const int bin = rot_bin(A.kf_kps[realIdxKF].angle, A.f_kps[bestIdxF].angle);
                                A.bin_f[A.kf_kf ? realIdxKF : bestIdxF] = (int8_t)bin;
                                atomicAdd(&A.histo[bin], 1);
                            }
                        }
                    }
                }
            }
            continue;
        }
        for (int iKF = A.kf_off[a]; iKF < A.kf_off[a + 1]; iKF++) {
            const int realIdxKF = A.kf_idx[iKF];
            if (!A.kf_has_mp[realIdxKF]) continue;
            const uint64_t* dq = (const uint64_t*)(A.kf_desc + (size_t)realIdxKF * 32);
            const uint64_t q0 = dq[0], q1 = dq[1], q2 = dq[2], q3 = dq[3];
            // candidate order = vector order (iF): key = dist << 32 | iF keeps the reference's first-wins tie rule
            uint64_t k0 = ~0ull, k1 = ~0ull;
            for (int iF = f0 + lane; iF < f1; iF += 64) {
                const int realIdxF = A.f_idx[iF];
                if (__hip_atomic_load(&A.match_f[realIdxF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) continue;   // vpMapPointMatches[realIdxF] / vbMatched2
                if (A.kf_kf && !A.f_has_mp[realIdxF]) continue;                      // !pMP2 || pMP2->isBad()
                const uint64_t* tp = (const uint64_t*)(A.f_desc + (size_t)realIdxF * 32);
                const int dist = __popcll(q0 ^ tp[0]) + __popcll(q1 ^ tp[1]) + __popcll(q2 ^ tp[2]) + __popcll(q3 ^ tp[3]);
                const uint64_t key = ((uint64_t)dist << 32) | (uint32_t)iF;
                if (key < k0) { k1 = k0; k0 = key; }
                else if (key < k1) k1 = key;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint64_t o0 = __shfl_xor(k0, d, 64), o1 = __shfl_xor(k1, d, 64);
                const uint64_t l0 = k0 < o0 ? k0 : o0, h0 = k0 < o0 ? o0 : k0, s1 = k1 < o1 ? k1 : o1;
                k0 = l0; k1 = h0 < s1 ? h0 : s1;
            }
            if (k0 != ~0ull && (int)(k0 >> 32) < 256) {
                const int bestDist1 = (int)(k0 >> 32);
                const int bestDist2 = (k1 != ~0ull && (int)(k1 >> 32) < 256) ? (int)(k1 >> 32) : 256;
                if ((A.kf_kf ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW) && (float)bestDist1 < A.nnratio * (float)bestDist2) {
                    if (lane == 0) {
                        const int bestIdxF = A.f_idx[(int)(k0 & 0xffffffffu)];
                        __hip_atomic_store(&A.match_f[bestIdxF], realIdxKF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        atomicAdd(A.nmatches, 1);
                        if (A.kf_kf) A.match12[realIdxKF] = bestIdxF;
                        if (A.checkOri) {
                            const int bin = rot_bin(A.kf_kps[realIdxKF].angle, A.f_kps[bestIdxF].angle);
                            A.bin_f[A.kf_kf ? realIdxKF : bestIdxF] = (int8_t)bin;      // rotHist holds idx1 (KF,KF) or bestIdxF (KF,F)
                            atomicAdd(&A.histo[bin], 1);
                        }
                    }
                    __threadfence();          // the next KeyFrame feature of this node must see match_f
                }
            }
        }
    }
}

__global__ void search_bow_finish_kernel(BowArgs A)
{
    __shared__ int keep;
    if (threadIdx.x == 0) {
        int h[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) h[i] = A.histo[i];
        int i1, i2, i3;
        three_maxima(h, HISTO_LENGTH, i1, i2, i3);
        int k = 0;
        if (i1 >= 0) k |= 1 << i1; if (i2 >= 0) k |= 1 << i2; if (i3 >= 0) k |= 1 << i3;
        keep = k;
    }
    __syncthreads();
    int dec = 0;
    const int nout = A.kf_kf ? A.n_kf : A.n_f;
    int32_t* out = A.kf_kf ? A.match12 : A.match_f;
    for (int i = threadIdx.x; i < nout; i += blockDim.x) {
        const int b = A.bin_f[i];
        if (b >= 0 && !(keep & (1 << b)) && out[i] >= 0) { out[i] = -1; dec++; }
    }
    if (dec) atomicSub(A.nmatches, dec);
}

__global__ void bow_init_kernel(int32_t* match_f, int8_t* bin_f, int n_f, int32_t* histo, int32_t* nmatches, int32_t* match12, int n_kf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_f) match_f[i] = -1;
    if (i < max(n_f, n_kf)) bin_f[i] = -1;
    if (match12 && i < n_kf) match12[i] = -1;
    if (i < 32) histo[i] = 0;
    if (i == 0) *nmatches = 0;
}

int search_bow_dev(eorb_ctx* c, const eorb_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_has_mp,
                   const uint32_t* kf_nodes, const int32_t* kf_off, const int32_t* kf_idx, int kf_nn,
                   const eorb_keypoint* f_kps, int n_f, const uint8_t* f_desc, const uint32_t* f_nodes, const int32_t* f_off,
                   const int32_t* f_idx, int f_nn, int32_t* match_f, int8_t* bin_f, int32_t* histo, int32_t* nmatches,
                   float nnratio, int checkOri, int kf_kf, const uint8_t* f_has_mp, int32_t* match12, int n_kf)
{
    BowArgs A{kf_kps, kf_desc, kf_has_mp, kf_nodes, kf_off, kf_idx, kf_nn, f_kps, n_f, f_desc, f_nodes, f_off, f_idx, f_nn,
              match_f, bin_f, histo, nmatches, nnratio, checkOri, kf_kf, f_has_mp, match12, n_kf};
    ProfScope ps(c, "search_bow");
    bow_init_kernel<<<(std::max(std::max(n_f, n_kf), 32) + 255) / 256, 256, 0, c->stream>>>(match_f, bin_f, n_f, histo, nmatches,
                                                                                          kf_kf ? match12 : nullptr, n_kf);
    if (kf_nn > 0 && f_nn > 0) {
        const int blocks = std::min((kf_nn + 3) / 4, 1024);
        search_bow_kernel<<<blocks, 256, 0, c->stream>>>(A);
        if (checkOri) search_bow_finish_kernel<<<1, 256, 0, c->stream>>>(A);
    }
    EORB_LAUNCH_CHECK(c, "search_bow kernels");
    return EORB_OK;
}

// ---------------------------------------------------------------------------------------------------
// local-mapping matcher: mono branch of ORBmatcher::SearchForTriangulation (:975-1214; MixedMatcher.cpp:1326-1573).
// vbMatched2 is never written in the reference, so every pKF1 feature is independent: one wave per feature-vector entry of
// pKF1, lanes over the pKF2 features of the same vocabulary node.  The sequential update rule (skip dist > bestDist, replace
// on pass) keeps the passing candidate with the least distance, the LAST one among equals: key = dist << 32 | ~position.

__device__ __forceinline__ bool epipolar_ok(float x1, float y1, float x2, float y2, const float* F, float unc)
{   // Pinhole::epipolarConstrain (src/CameraModels/Pinhole.cpp:142-156) with F12 given
    const float a = x1 * F[0] + y1 * F[3] + F[6];
    const float b = x1 * F[1] + y1 * F[4] + F[7];
    const float c = x1 * F[2] + y1 * F[5] + F[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84f * unc;
}

__global__ __launch_bounds__(256) void search_tri_kernel(TriArgs A)
{
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const int total = A.off1[A.nn1];
    for (int p = gw; p < total; p += nw) {
        const int id1 = A.idx1[p];
        const uint8_t e1 = A.elig1[id1];                // bit 0: eligible; bit 1: rectified-stereo keypoint (mvuRight >= 0, bStereo1 :1051)
        if (!(e1 & 1)) continue;
        int lo = 0, hi = A.nn1;                         // node a with off1[a] <= p < off1[a+1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.off1[mid] <= p) lo = mid; else hi = mid; }
        const uint32_t node = A.nodes1[lo];
        int l2 = 0, h2 = A.nn2;
        while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (A.nodes2[mid] < node) l2 = mid + 1; else h2 = mid; }
        if (l2 >= A.nn2 || A.nodes2[l2] != node) continue;
        const eorb_keypoint kp1 = A.kps1[id1];
        uint64_t q0, q1, q2, q3;
        load_desc32(A.desc1 + (size_t)id1 * A.stride1, q0, q1, q2, q3);
        uint64_t k0 = ~0ull;
        for (int i2 = A.off2[l2] + lane; i2 < A.off2[l2 + 1]; i2 += 64) {
            const int id2 = A.idx2[i2];
            const uint8_t e2 = A.elig2[id2];
            if (!(e2 & 1)) continue;
            uint64_t t0, t1, t2, t3;
            load_desc32(A.desc2 + (size_t)id2 * A.stride2, t0, t1, t2, t3);
            const int dist = __popcll(q0 ^ t0) + __popcll(q1 ^ t1) + __popcll(q2 ^ t2) + __popcll(q3 ^ t3);
            if (dist > TH_LOW) continue;
            const eorb_keypoint kp2 = A.kps2[id2];
            if (kp2.octave < 0 || kp2.octave >= A.nlevels) continue;            // rejected on the host already
            if (!((e1 | e2) & 2)) {                                            // "if(!bStereo1 && !bStereo2 && !pKF1->mpCamera2)" :1093
                const float distex = A.epx - kp2.x, distey = A.epy - kp2.y;
                if (distex * distex + distey * distey < 100 * A.scale2[kp2.octave]) continue;
            }
            if (!(A.bCoarse || epipolar_ok(kp1.x, kp1.y, kp2.x, kp2.y, A.F, A.sigma2_2[kp2.octave]))) continue;
            const uint64_t key = ((uint64_t)dist << 32) | (uint32_t)(~(uint32_t)i2);
            if (key < k0) k0 = key;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = __shfl_xor(k0, d, 64); k0 = o < k0 ? o : k0; }
        if (lane == 0 && k0 != ~0ull) {
            const int bestIdx2 = A.idx2[(int)(~(uint32_t)(k0 & 0xffffffffu))];
            A.match12[id1] = bestIdx2;
            atomicAdd(A.nmatches, 1);
            if (A.checkOri) {
                const int bin = rot_bin(kp1.angle, A.kps2[bestIdx2].angle);
                A.bin1[id1] = (int8_t)bin;
                atomicAdd(&A.histo[bin], 1);
            }
        }
    }
}

int search_tri_dev(eorb_ctx* c, const TriArgs& A)
{
    ProfScope ps(c, "search_for_triangulation");
    bow_init_kernel<<<(std::max(A.n1, 32) + 255) / 256, 256, 0, c->stream>>>(A.match12, A.bin1, A.n1, A.histo, A.nmatches, nullptr, 0);
    search_tri_kernel<<<std::min((A.n1 + 3) / 4 + 1, 2048), 256, 0, c->stream>>>(A);
    if (A.checkOri) {
        BowArgs B{};
        B.kf_kf = 1; B.n_kf = A.n1; B.match12 = A.match12; B.bin_f = A.bin1; B.histo = A.histo; B.nmatches = A.nmatches;
        search_bow_finish_kernel<<<1, 256, 0, c->stream>>>(B);
    }
    EORB_LAUNCH_CHECK(c, "search_for_triangulation kernels");
    return EORB_OK;
}

// ---------------------------------------------------------------------------------------------------
// search core of ORBmatcher::Fuse (:1512-1578, :1700-1720), SearchBySim3 (:1829-1860, :1909-1940) and
// SearchByProjection(KeyFrame*, Scw, ...) (:548-588): best keypoint (least distance, first in GetFeaturesInArea order) with
// octave in [L-1, L] inside the radius.  SEQ = false: queries independent, one wave each.  SEQ = true (vpMatched feeds later
// queries): one workgroup walks the queries in order.

__global__ void kf_cells_kernel(const eorb_keypoint* __restrict__ kps, int n, GridB g, uint16_t* __restrict__ cell)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int px = (int)roundf((kps[i].x - g.minX) * g.invW);
    const int py = (int)roundf((kps[i].y - g.minY) * g.invH);
    cell[i] = (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) ? (uint16_t)0xFFFF : (uint16_t)(px * kGridRows + py);
}

template <bool SEQ>
__device__ __forceinline__ uint64_t radius_scan(const RadArgs& A, int m, int first, int step, const uint8_t* taken)
{
    const float u = A.uv[2 * m], v = A.uv[2 * m + 1], r = A.radius[m];
    int cx0, cx1, cy0, cy1;
    uint64_t k0 = ~0ull;
    if (!cell_range(A.g, u, v, r, cx0, cx1, cy0, cy1)) return k0;
    const int L = A.level[m];
    uint64_t q0, q1, q2, q3;
    load_desc32(A.q_desc + (size_t)m * 32, q0, q1, q2, q3);
    for (int i = first; i < A.n; i += step) {
        const int cell = A.cell[i];
        if (cell == 0xFFFF) continue;
        const int cx = cell / kGridRows, cy = cell - cx * kGridRows;
        if (cx < cx0 || cx > cx1 || cy < cy0 || cy > cy1) continue;
        const eorb_keypoint k = A.kps[i];
        const float distx = k.x - u, disty = k.y - v;
        if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
        if (SEQ && taken[i]) continue;
        if (k.octave < L - 1 || k.octave > L) continue;
        if (A.inv_sigma2) {
            if (k.octave < 0 || k.octave >= A.nlevels) continue;
            const float ex = u - k.x, ey = v - k.y;
            const float kpr = A.uright ? A.uright[i] : -1.f;
            if (kpr >= 0.f) {                                                  // "Check reprojection error in stereo" :1541-1553
                const float er = A.q_ur[m] - kpr;
                const float e2 = ex * ex + ey * ey + er * er;
                if ((double)(e2 * A.inv_sigma2[k.octave]) > 7.8) continue;
            } else {
                const float e2 = ex * ex + ey * ey;
                if ((double)(e2 * A.inv_sigma2[k.octave]) > 5.99) continue;
            }
        }
        uint64_t t0, t1, t2, t3;
        load_desc32(A.desc + (size_t)i * A.stride, t0, t1, t2, t3);
        const int dist = __popcll(q0 ^ t0) + __popcll(q1 ^ t1) + __popcll(q2 ^ t2) + __popcll(q3 ^ t3);
        const uint64_t key = ((uint64_t)dist << 44) | ((uint64_t)cell << 32) | (uint32_t)i;
        if (key < k0) k0 = key;
    }
    return k0;
}

__global__ __launch_bounds__(256) void kf_radius_kernel(RadArgs A)
{
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (int m = gw; m < A.M; m += nw) {
        uint64_t k0 = ~0ull;
        if (A.valid[m]) k0 = radius_scan<false>(A, m, lane, 64, nullptr);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = __shfl_xor(k0, d, 64); k0 = o < k0 ? o : k0; }
        if (lane == 0) {
            const bool ok = k0 != ~0ull && (int)(k0 >> 44) < 256;
            A.best_idx[m] = ok ? (int)(k0 & 0xffffffffu) : -1;
            A.best_dist[m] = ok ? (int)(k0 >> 44) : 256;
        }
    }
}

__global__ __launch_bounds__(256) void kf_radius_seq_kernel(RadArgs A)
{
    extern __shared__ unsigned char smem[];
    uint64_t* red = (uint64_t*)smem;
    uint8_t* taken = (uint8_t*)(red + 8);
    for (int i = threadIdx.x; i < A.n; i += blockDim.x) taken[i] = A.taken[i];
    __syncthreads();
    for (int m = 0; m < A.M; m++) {
        uint64_t k0 = ~0ull, k1 = ~0ull;
        if (A.valid[m]) k0 = radius_scan<true>(A, m, threadIdx.x, blockDim.x, taken);
        block_top2(k0, k1, red);
        if (threadIdx.x == 0) {
            const bool ok = k0 != ~0ull && (int)(k0 >> 44) < 256;
            const int bi = ok ? (int)(k0 & 0xffffffffu) : -1, bd = ok ? (int)(k0 >> 44) : 256;
            A.best_idx[m] = bi; A.best_dist[m] = bd;
            if (ok && (float)bd <= A.accept_thr) taken[bi] = 1;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < A.n; i += blockDim.x) A.taken[i] = taken[i];
}

int kf_radius_dev(eorb_ctx* c, const RadArgs& A, uint16_t* d_cell)
{
    if (A.M <= 0) return EORB_OK;
    ProfScope ps(c, "kf_radius_match");
    if (A.n > 0) kf_cells_kernel<<<(A.n + 255) / 256, 256, 0, c->stream>>>(A.kps, A.n, A.g, d_cell);
    if (A.taken) {
        const size_t lds = 64 + (((size_t)A.n + 15) & ~(size_t)15);
        if (lds > 160 * 1024) return set_err(c, EORB_E_CAPACITY, "kf_radius_match: %d keypoints exceed the LDS flags", A.n);
        hipFuncSetAttribute((const void*)kf_radius_seq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        kf_radius_seq_kernel<<<1, 256, lds, c->stream>>>(A);
    } else {
        kf_radius_kernel<<<std::min((A.M + 3) / 4, 4096), 256, 0, c->stream>>>(A);
    }
    EORB_LAUNCH_CHECK(c, "kf_radius_match kernels");
    return EORB_OK;
}

// ---------------------------------------------------------------------------------------------------
// DBoW2 TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1125-1250): the producer of the feature vectors the SearchByBoW kernels consume.
// K-a: one wave per feature walks the tree; lanes = children of the current node (FORB::distance = popcount), first minimum wins.
__global__ __launch_bounds__(256) void bow_descend_kernel(const uint8_t* __restrict__ desc, int n, int stride, BowVoc V, int nid_level,
                                                          uint32_t* __restrict__ word_of, double* __restrict__ w_of, uint32_t* __restrict__ node_of)
{
    const int lane = threadIdx.x & 63;
    const int f = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (f >= n) return;
    uint64_t q0, q1, q2, q3;
    load_desc32(desc + (size_t)f * stride, q0, q1, q2, q3);
    int node = 0, level = 0, nid = 0;
    do {
        ++level;
        const int c0 = V.child_off[node], nc = V.child_off[node + 1] - c0;
        uint64_t key = ~0ull;
        for (int c = lane; c < nc; c += 64) {
            const int id = V.child_ids[c0 + c];
            uint64_t t0, t1, t2, t3;
            load_desc32(V.node_desc + (size_t)id * 32, t0, t1, t2, t3);
            const int d = __popcll(q0 ^ t0) + __popcll(q1 ^ t1) + __popcll(q2 ^ t2) + __popcll(q3 ^ t3);
            const uint64_t k = ((uint64_t)d << 32) | (uint32_t)c;
            key = k < key ? k : key;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = __shfl_xor(key, d, 64); key = o < key ? o : key; }
        node = V.child_ids[c0 + (int)(key & 0xffffffffu)];
        if (level == nid_level) nid = node;
    } while (V.child_off[node + 1] > V.child_off[node] && level < 64);
    if (lane == 0) {
        const double w = V.weight[node];
        const bool keep = w > 0;                               // stopped words carry weight 0
        word_of[f] = keep ? (uint32_t)V.word_id[node] : 0xffffffffu;
        w_of[f] = w;
        node_of[f] = keep ? (uint32_t)nid : 0xffffffffu;
    }
}

// bitonic sort of P (power of two) 64-bit keys in LDS by the whole workgroup (1024 threads).  Thread t keeps keys t, t + 1024, ... in
// registers: partners less than 64 apart are exchanged by lane shuffles, partners a multiple of 1024 apart are the thread's own
// registers, and only the distances 64 .. 512 go through LDS and two barriers (10 of the 55 steps of a 1024-key sort).
__device__ __forceinline__ uint64_t blk_shfl_xor_u64(uint64_t v, int j)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, j, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), j, 64);
    return ((uint64_t)hi << 32) | lo;
}
template <int M>
__device__ void block_bitonic_sort_regs(uint64_t* keys, int P)
{
    constexpr int NT = 1024;
    const int tid = threadIdx.x;
    __syncthreads();
    uint64_t v[M];
#pragma unroll
    for (int m = 0; m < M; m++) { const int i = tid + NT * m; v[m] = i < P ? keys[i] : ~0ull; }
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= NT) {
#pragma unroll
                for (int dm = 1; dm < M; dm <<= 1) {
                    if (j != dm * NT) continue;
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        if ((m & dm) == 0 && (m | dm) < M) {
                            const int i = tid + NT * m;
                            const bool up = (i & k) == 0;
                            const uint64_t x = v[m], y = v[m | dm];
                            const bool sw = (x > y) == up;
                            v[m] = sw ? y : x; v[m | dm] = sw ? x : y;
                        }
                    }
                }
            } else if (j >= 64) {
#pragma unroll
                for (int m = 0; m < M; m++) { const int i = tid + NT * m; if (i < P) keys[i] = v[m]; }
                __syncthreads();
                uint64_t o[M];
#pragma unroll
                for (int m = 0; m < M; m++) { const int i = tid + NT * m; o[m] = i < P ? keys[i ^ j] : ~0ull; }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const int i = tid + NT * m;
                    const bool up = (i & k) == 0, low = (i & j) == 0;
                    const uint64_t mn = v[m] < o[m] ? v[m] : o[m], mx = v[m] < o[m] ? o[m] : v[m];
                    v[m] = (low == up) ? mn : mx;
                }
            } else {
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const int i = tid + NT * m;
                    const uint64_t o = blk_shfl_xor_u64(v[m], j);
                    const bool up = (i & k) == 0, low = (i & j) == 0;
                    const uint64_t mn = v[m] < o ? v[m] : o, mx = v[m] < o ? o : v[m];
                    v[m] = (low == up) ? mn : mx;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < M; m++) { const int i = tid + NT * m; if (i < P) keys[i] = v[m]; }
    __syncthreads();
}
__device__ __forceinline__ void block_bitonic_sort(uint64_t* keys, int P)
{
    if (blockDim.x == 1024) {
        if (P <= 1024) { block_bitonic_sort_regs<1>(keys, P); return; }
        if (P <= 2048) { block_bitonic_sort_regs<2>(keys, P); return; }
        if (P <= 4096) { block_bitonic_sort_regs<4>(keys, P); return; }
        if (P <= 8192) { block_bitonic_sort_regs<8>(keys, P); return; }
    }
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < P; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
        }
    }
    __syncthreads();
}

// run heads of the sorted keys (upper 32 bits = run id) -> exclusive run index per element, number of runs; wave 0 scans
__device__ __forceinline__ int block_run_index(const uint64_t* keys, int m, uint32_t* ridx)
{
    __shared__ int s_runs;
    if (threadIdx.x < 64) {
        int run = 0;
        for (int i0 = 0; i0 < m; i0 += 64) {
            const int i = i0 + (int)threadIdx.x;
            const int head = (i < m) && (i == 0 || (uint32_t)(keys[i] >> 32) != (uint32_t)(keys[i - 1] >> 32));
            int incl = head;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if ((int)threadIdx.x >= d) incl += t; }
            if (i < m) ridx[i] = (uint32_t)(run + incl - 1);
            run += __shfl(incl, 63, 64);
        }
        if (threadIdx.x == 0) s_runs = run;
    }
    __syncthreads();
    return s_runs;
}

// K-b: one workgroup turns the per-feature (word, weight, node) triples into the BowVector (std::map order = ascending word id,
// weights accumulated in feature order, then BowVector::normalize) and the FeatureVector (ascending node id, features in
// push_back order).
__global__ __launch_bounds__(1024) void bow_assemble_kernel(const uint32_t* __restrict__ word_of, const double* __restrict__ w_of,
                                                            const uint32_t* __restrict__ node_of, int n, int P, int weighting, int norm,
                                                            uint32_t* __restrict__ bow_word, double* __restrict__ bow_val,
                                                            uint32_t* __restrict__ fv_node, int32_t* __restrict__ fv_off, int32_t* __restrict__ fv_idx,
                                                            int32_t* __restrict__ counts /* n_words, n_fvnodes */)
{
    extern __shared__ unsigned char smem[];
    uint64_t* keys = (uint64_t*)smem;                       // P
    double* vals = (double*)(keys + P);                     // P: the words' values (summed for the norm from here, not from global memory)
    uint32_t* ridx = (uint32_t*)(vals + P);                 // P
    __shared__ int s_m;
    __shared__ double s_norm;
    // ---- BowVector ----
    for (int i = threadIdx.x; i < P; i += blockDim.x)
        keys[i] = (i < n && word_of[i] != 0xffffffffu) ? (((uint64_t)word_of[i] << 32) | (uint32_t)i) : ~0ull;
    block_bitonic_sort(keys, P);
    if (threadIdx.x == 0) {                                  // valid features sort first: m = first sentinel
        int lo = 0, hi = P;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] != ~0ull) lo = mid + 1; else hi = mid; }
        s_m = lo;
    }
    __syncthreads();
    const int m = s_m;
    const int nw = block_run_index(keys, m, ridx);
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        if (i == 0 || ridx[i] != ridx[i - 1]) {              // run head: addWeight in feature order (same word -> same weight)
            const double w = w_of[(int)(keys[i] & 0xffffffffu)];
            double v = w;
            if (weighting == 0 || weighting == 1)
                for (int j = i + 1; j < m && ridx[j] == ridx[i]; j++) v += w;
            bow_word[ridx[i]] = (uint32_t)(keys[i] >> 32);
            vals[ridx[i]] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double nrm = 0.0;
        if (norm == 1) { for (int i = 0; i < nw; i++) nrm += fabs(vals[i]); }
        else if (norm == 2) { for (int i = 0; i < nw; i++) nrm += vals[i] * vals[i]; nrm = sqrt(nrm); }
        else if (weighting == 0 || weighting == 1) nrm = (double)nw;     // "unnecessary when normalizing" branch :1166-1172
        s_norm = nrm;
        counts[0] = nw;
    }
    __syncthreads();
    {
        const bool div = s_norm > 0.0 && (norm != 0 || weighting == 0 || weighting == 1);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) bow_val[i] = div ? vals[i] / s_norm : vals[i];
    }
    __syncthreads();
    // ---- FeatureVector ----
    for (int i = threadIdx.x; i < P; i += blockDim.x)
        keys[i] = (i < n && node_of[i] != 0xffffffffu) ? (((uint64_t)node_of[i] << 32) | (uint32_t)i) : ~0ull;
    block_bitonic_sort(keys, P);
    const int nn = block_run_index(keys, m, ridx);
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        fv_idx[i] = (int32_t)(keys[i] & 0xffffffffu);
        if (i == 0 || ridx[i] != ridx[i - 1]) { fv_node[ridx[i]] = (uint32_t)(keys[i] >> 32); fv_off[ridx[i]] = i; }
    }
    if (threadIdx.x == 0) { fv_off[nn] = m; counts[1] = nn; }
}

int bow_transform_dev(eorb_ctx* c, const uint8_t* d_desc, int n, int stride, const BowVoc& V, int levelsup, int weighting, int norm,
                      uint32_t* d_word_of, double* d_w_of, uint32_t* d_node_of, uint32_t* d_bow_word, double* d_bow_val,
                      uint32_t* d_fv_node, int32_t* d_fv_off, int32_t* d_fv_idx, int32_t* d_counts)
{
    int P = 64; while (P < n) P <<= 1;
    const size_t lds = (size_t)P * 20;
    if (lds > 150 * 1024) return set_err(c, EORB_E_CAPACITY, "bow_transform: %d features exceed the LDS sort", n);
    ProfScope ps(c, "bow_transform");
    bow_descend_kernel<<<(n + 3) / 4, 256, 0, c->stream>>>(d_desc, n, stride, V, V.L - levelsup, d_word_of, d_w_of, d_node_of);
    hipFuncSetAttribute((const void*)bow_assemble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    bow_assemble_kernel<<<1, 1024, lds, c->stream>>>(d_word_of, d_w_of, d_node_of, n, P, weighting, norm, d_bow_word, d_bow_val,
                                                     d_fv_node, d_fv_off, d_fv_idx, d_counts);
    EORB_LAUNCH_CHECK(c, "bow_transform kernels");
    return EORB_OK;
}

// generic windowed matcher (SURVEY §8(b) eorb_hamming_window_match): one wave per query, lanes over its candidate list; key =
// dist << 32 | position keeps the reference's first-wins order for both the best and the second best
__global__ __launch_bounds__(256) void window_match_kernel(const uint8_t* __restrict__ q_desc, int nq, int q_stride,
                                                           const uint8_t* __restrict__ t_desc, int t_stride,
                                                           const int32_t* __restrict__ off, const int32_t* __restrict__ cand,
                                                           int32_t* __restrict__ out /* 4 x nq */)
{
    const int lane = threadIdx.x & 63;
    const int q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (q >= nq) return;
    uint64_t q0, q1, q2, q3;
    load_desc32(q_desc + (size_t)q * q_stride, q0, q1, q2, q3);
    uint64_t k0 = ~0ull, k1 = ~0ull;
    const int c0 = off[q], c1 = off[q + 1];
    for (int k = c0 + lane; k < c1; k += 64) {
        const int c = cand[k];
        uint64_t t0, t1, t2, t3;
        load_desc32(t_desc + (size_t)c * t_stride, t0, t1, t2, t3);
        const int d = __popcll(q0 ^ t0) + __popcll(q1 ^ t1) + __popcll(q2 ^ t2) + __popcll(q3 ^ t3);
        const uint64_t key = ((uint64_t)d << 32) | (uint32_t)(k - c0);
        if (key < k0) { k1 = k0; k0 = key; } else if (key < k1) k1 = key;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t o0 = __shfl_xor(k0, d, 64), o1 = __shfl_xor(k1, d, 64);
        const uint64_t lo = k0 < o0 ? k0 : o0, hi = k0 < o0 ? o0 : k0, s1 = k1 < o1 ? k1 : o1;
        k0 = lo; k1 = hi < s1 ? hi : s1;
    }
    if (lane == 0) {
        // the sequential rule only records a second best that was seen while it was >= the then-best: with first-wins ties the
        // top two keys of (dist, position) are exactly best and second
        const bool hb = k0 != ~0ull, hs = k1 != ~0ull;
        out[q] = hb ? cand[c0 + (int)(k0 & 0xffffffffu)] : -1;
        out[nq + q] = hb ? (int)(k0 >> 32) : 256;
        out[2 * nq + q] = hs ? cand[c0 + (int)(k1 & 0xffffffffu)] : -1;
        out[3 * nq + q] = hs ? (int)(k1 >> 32) : 256;
    }
}

int window_match_dev(eorb_ctx* c, const uint8_t* d_q, int nq, int q_stride, const uint8_t* d_t, int t_stride, const int32_t* d_off,
                     const int32_t* d_cand, int32_t* d_out)
{
    if (nq <= 0) return EORB_OK;
    ProfScope ps(c, "hamming_window_match");
    window_match_kernel<<<(nq + 3) / 4, 256, 0, c->stream>>>(d_q, nq, q_stride, d_t, t_stride, d_off, d_cand, d_out);
    EORB_LAUNCH_CHECK(c, "window_match_kernel");
    return EORB_OK;
}

// MixedFrame::sortFeaturesResponse (MixedFrame.cpp:211-225): stable descending order by response.
// rank(i) = #{j : r_j > r_i} + #{j < i : r_j == r_i}; perm[rank(i)] = i.  n is a few thousand at most.
__global__ void sort_response_kernel(const eorb_keypoint* __restrict__ kps, int n, int32_t* __restrict__ perm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = kps[i].response;
    int rank = 0;
    for (int j = 0; j < n; j++) {
        const float q = kps[j].response;
        rank += (q > r) || (q == r && j < i);
    }
    perm[rank] = i;
}

int sort_response_dev(eorb_ctx* c, const eorb_keypoint* d_kps, int n, int32_t* d_perm)
{
    if (n <= 0) return EORB_OK;
    ProfScope ps(c, "sort_response");
    sort_response_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(d_kps, n, d_perm);
    EORB_LAUNCH_CHECK(c, "sort_response_kernel");
    return EORB_OK;
}


// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:349-423), batched over map points: one workgroup per map point,
// thread i owns row i of the N x N distance matrix; its median is read off a 257-bin counting histogram (distances are
// integers in [0, 256]); the first row with the smallest median wins.
__global__ __launch_bounds__(256) void distinctive_kernel(const uint8_t* __restrict__ desc, const int32_t* __restrict__ offsets,
                                                          int32_t* __restrict__ best)
{
    __shared__ uint16_t hist[64 * 260];
    __shared__ unsigned long long s_best;
    const int m = blockIdx.x;
    const int o0 = offsets[m], N = offsets[m + 1] - o0;
    if (N <= 0) { if (threadIdx.x == 0) best[m] = -1; return; }
    if (threadIdx.x == 0) s_best = ~0ull;
    __syncthreads();
    const uint8_t* D = desc + 32 * (size_t)o0;
    const int k = (int)(0.5 * (double)(N - 1));          // vDists[0.5*(N-1)]
    for (int r0 = 0; r0 < N; r0 += 64) {                  // 64 rows at a time; 4 threads per row split the columns
        const int row = r0 + (threadIdx.x >> 2), part = threadIdx.x & 3;
        uint16_t* h = hist + (threadIdx.x >> 2) * 260;
        for (int b = part; b < 257; b += 4) h[b] = 0;
        __syncthreads();
        if (row < N) {
            uint64_t q[4];
            load_desc32(D + 32 * (size_t)row, q[0], q[1], q[2], q[3]);
            for (int j = part; j < N; j += 4) {
                uint64_t t0, t1, t2, t3;
                load_desc32(D + 32 * (size_t)j, t0, t1, t2, t3);
                const int d = (j == row) ? 0 : (__popcll(q[0] ^ t0) + __popcll(q[1] ^ t1) + __popcll(q[2] ^ t2) + __popcll(q[3] ^ t3));
                atomicAdd((unsigned int*)(h + (d & ~1)), (d & 1) ? 0x10000u : 1u);     // 16-bit counter inside its 32-bit word
            }
        }
        __syncthreads();
        if (row < N && part == 0) {
            int acc = 0, median = 256;
            for (int b = 0; b < 257; b++) { acc += h[b]; if (acc > k) { median = b; break; } }
            const unsigned long long key = ((unsigned long long)median << 32) | (unsigned int)row;   // first minimum wins
            atomicMin(&s_best, key);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) best[m] = (int32_t)(s_best & 0xffffffffu);
}

int distinctive_dev(eorb_ctx* c, const uint8_t* d_desc, const int32_t* d_offsets, int M, int32_t* d_best)
{
    if (M <= 0) return EORB_OK;
    ProfScope ps(c, "distinctive_descriptors");
    distinctive_kernel<<<M, 256, 0, c->stream>>>(d_desc, d_offsets, d_best);
    EORB_LAUNCH_CHECK(c, "distinctive_kernel");
    return EORB_OK;
}

int bf_knn2_dev(eorb_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx2, int32_t* d_dist2)
{
    if (nq <= 0) return EORB_OK;
    ProfScope ps(c, "bf_knn2");
    if (nq >= 16 * 1024) bf_knn2_kernel<16><<<(nq + 15) / 16, 256, 0, c->stream>>>(d_q, nq, d_t, nt, d_idx2, d_dist2);
    else if (nq >= 8 * 512) bf_knn2_kernel<32><<<(nq + 7) / 8, 256, 0, c->stream>>>(d_q, nq, d_t, nt, d_idx2, d_dist2);
    else bf_knn2_kernel<64><<<(nq + 3) / 4, 256, 0, c->stream>>>(d_q, nq, d_t, nt, d_idx2, d_dist2);
    EORB_LAUNCH_CHECK(c, "bf_knn2_kernel");
    return EORB_OK;
}

}  // namespace eorb
