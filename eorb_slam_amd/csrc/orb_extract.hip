// orb_extract.hip -- ORB extraction on gfx950, bit-exact w.r.t. the reference's CPU ORBextractor.
//
// Replaces ORBextractor::operator() (src/ORBextractor.cc:1092-1238 of the reference) and everything
// under it.  All kernels are batched over time-slices (frames); one launch covers every slice and,
// where the data dependences allow it, every pyramid level.
//   pyr_level0_kernel   copyMakeBorder(image, REFLECT_101)                       (:1258-1262)
//   pyr_resize_kernel   cv::resize INTER_LINEAR 8u (fixed-point, 11-bit coefficients) of level l-1 into
//                       level l + REFLECT_101 border, one launch per level         (:1250-1257)
//   fast_cells_kernel   one workgroup per (slice, level, cell): FAST-9/16 score of the cell's detection
//                       domain from an LDS-staged tile, 3x3 NMS, iniThFAST -> minThFAST fallback,
//                       raster-ordered emission                                    (:784-878, cv::FAST)
//   octree_kernel       DistributeOctTree (:558-782): the reference's sequential list algorithm, run by
//                       one wavefront per (slice, level); only DivideNode's key partition is lane-parallel
//   orient_kernel       IC_Angle (:77-104) + cv::fastAtan2, 32 lanes per keypoint (lane = column)
//   blur_kernel         GaussianBlur(5x5, sigma 2, REFLECT_101) 8u Q8 separable     (:1141-1142)
//   brief_kernel        computeOrbDescriptor (:108-157): 32 lanes per keypoint, 16 taps each
//   assemble_kernel     output ordering of operator() (:1150-1173): scale, lapping-area back-fill
#include "eorb_ctx.h"
#include "dev_math.h"
#include "orb_pattern.h"
#include <math.h>
#include <algorithm>
#include <string.h>
#include <vector>

namespace eorb {

// device copy of LevelGeom (trivially copyable); kept in constant-like global memory
struct DevGeom {
    LevelGeom lv[kMaxLevels];
    int nlevels, edge, W, H;
    int pyr_bytes, roi_bytes, ncells, cell_cap, cand_total, kp_total, max_out;
    int iniTh, minTh;
    int umax[16];
    float sf[kMaxLevels];
    // octree working set: items 0 the two node arrays, 1 / 2 the two (size, seq, node) lists, 3 / 4 the key ping-pong buffers, 5 the
    // candidate points, 6 the per-round arrays; each lives in LDS (offset into the dynamic LDS block) or, when the 160 KB do not hold
    // it, in the (slice, level) block of a global scratch buffer (offset into that block)
    // octree working set, two placements: [0] as much as the LDS holds (one workgroup per CU: single frames), [1] at most half of
    // it (two workgroups per CU: launches with more workgroups than CUs)
    int oct_in_lds[2][7], oct_off[2][7];
    int oct_lds_bytes[2], oct_gblock_bytes[2];
    // dynamic placement (levels whose CAPACITY does not fit the LDS but whose candidates of this frame do): the fixed items (6, 0, 1, 2)
    // at oct_off_dyn, the key buffers and the points behind them (from oct_dyn_base) sized by the level's candidates; 0: not in use
    int oct_dyn[2], oct_off_dyn[2][7], oct_dyn_base[2], oct_dyn_lds[2];
    int oct_direct_cap[2];       // most candidates of a level whose full passes are computed directly (0: never): the sort's buffer = the two size lists
    int node_cap_max, vsp_cap_max, ncap_max;
};

__device__ __constant__ signed char c_pattern[1024];

// ---------------------------------------------------------------------------------------------------
// pyramid
__global__ void pyr_level0_kernel(const uint8_t* __restrict__ img, int stride, size_t slice_bytes,
                                  const DevGeom* __restrict__ G, uint8_t* __restrict__ pyr, int32_t* __restrict__ err_flag)
{
    // (the first kernel of an extraction also clears its overflow flag: one launch less than a memset)
    if (err_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *err_flag = 0;
    const LevelGeom& L = G->lv[0];
    const int slice = blockIdx.y;
    const int n = L.bw * L.bh;
    uint8_t* dst = pyr + (size_t)slice * G->pyr_bytes + L.buf_off;
    const uint8_t* src = img + (size_t)slice * slice_bytes;
    // four destination bytes per thread (level buffers start on 64-byte boundaries; the tail of the last word is padding)
    for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 * 4 < n; i4 += gridDim.x * blockDim.x) {
        int y = (i4 * 4) / L.bw, x = i4 * 4 - y * L.bw;
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sy = reflect101(min(y, L.bh - 1) - G->edge, L.h), sx = reflect101(x - G->edge, L.w);
            w |= (uint32_t)src[(size_t)sy * stride + sx] << (8 * k);
            if (++x == L.bw) { x = 0; y++; }
        }
        ((uint32_t*)dst)[i4] = w;
    }
}

// The same with cv::normalize(MINMAX, 0..255, CV_8U) of the float event image folded in (EventConversion.cc:207-212; the
// arithmetic of ev_normalize_kernel): level 0 is built straight from the accumulated float image and its running extremes, the u8
// image -- an output of its own -- is written by the threads that hold its pixels.  One launch less on the event path.
__device__ __forceinline__ unsigned dec_mm_f32(unsigned e) { return (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e; }
__global__ void pyr_level0_f32_kernel(const float* __restrict__ f32, const uint32_t* __restrict__ mm, uint8_t* __restrict__ img_out, int stride,
                                      size_t slice_bytes, const DevGeom* __restrict__ G, uint8_t* __restrict__ pyr, int32_t* __restrict__ err_flag)
{
    if (err_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *err_flag = 0;
    const LevelGeom& L = G->lv[0];
    const int slice = blockIdx.y;
    const int n = L.bw * L.bh;
    uint8_t* dst = pyr + (size_t)slice * G->pyr_bytes + L.buf_off;
    const float* src = f32 + (size_t)slice * L.w * L.h;
    uint8_t* out = img_out + (size_t)slice * slice_bytes;
    const float mn = __uint_as_float(dec_mm_f32(mm[2 * slice])), mx = __uint_as_float(dec_mm_f32(mm[2 * slice + 1]));
    const float alpha = 255.f / (mx - mn);
    const float beta = -mn * alpha;
    for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 * 4 < n; i4 += gridDim.x * blockDim.x) {
        int y = (i4 * 4) / L.bw, x = i4 * 4 - y * L.bw;
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int yy = min(y, L.bh - 1) - G->edge, xx = x - G->edge;
            const int sy = reflect101(yy, L.h), sx = reflect101(xx, L.w);
            const float m = src[(size_t)sy * L.w + sx] * alpha;
            const float v = m + beta;
            int iv = __float2int_rn(v);
            iv = min(max(iv, 0), 255);
            w |= (uint32_t)iv << (8 * k);
            if (sy == yy && sx == xx && y < L.bh) out[(size_t)sy * stride + sx] = (uint8_t)iv;
            if (++x == L.bw) { x = 0; y++; }
        }
        ((uint32_t*)dst)[i4] = w;
    }
}

// xtab: per destination column {int16 sx, int16 a0, int16 a1, pad}; ytab: per row {sy0, sy1, b0, b1}
__global__ void pyr_resize_kernel(int level, const DevGeom* __restrict__ G, const short4* __restrict__ tabs,
                                  uint8_t* __restrict__ pyr)
{
    const LevelGeom& L = G->lv[level];
    const LevelGeom& P = G->lv[level - 1];
    const int slice = blockIdx.y;
    uint8_t* base = pyr + (size_t)slice * G->pyr_bytes;
    const uint8_t* sroi = base + P.buf_off + (size_t)G->edge * P.bw + G->edge;
    uint8_t* dst = base + L.buf_off;
    const short4* xt = tabs + L.xtab_off;
    const short4* yt = tabs + L.ytab_off;
    const int n = L.bw * L.bh;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = i / L.bw, x = i - y * L.bw;
        const int dy = reflect101(y - G->edge, L.h), dx = reflect101(x - G->edge, L.w);
        const short4 xc = xt[dx], yc = yt[dy];
        const uint8_t* S0 = sroi + (size_t)yc.x * P.bw;
        const uint8_t* S1 = sroi + (size_t)yc.y * P.bw;
        int r0, r1;
        if (dx < L.xmax) {
            r0 = S0[xc.x] * xc.y + S0[xc.x + 1] * xc.z;
            r1 = S1[xc.x] * xc.y + S1[xc.x + 1] * xc.z;
        } else {
            r0 = S0[xc.x] * 2048;
            r1 = S1[xc.x] * 2048;
        }
        dst[i] = (uint8_t)((((yc.z * (r0 >> 4)) >> 16) + ((yc.w * (r1 >> 4)) >> 16) + 2) >> 2);
    }
}

// ---------------------------------------------------------------------------------------------------
// FAST-9/16 score: S = max over the 16 contiguous 9-arcs of min(v - p) (darker) / min(p - v) (brighter).
// A pixel is a corner at threshold t iff S > t and its cv::FAST score (cornerScore<16>) is S - 1.
__device__ __forceinline__ int fast_S(const uint8_t* __restrict__ t, int pitch)
{
    const int v = t[0];
    int d[16];
    d[0] = v - t[3 * pitch];          d[1] = v - t[3 * pitch + 1];   d[2] = v - t[2 * pitch + 2];   d[3] = v - t[pitch + 3];
    d[4] = v - t[3];                  d[5] = v - t[-pitch + 3];      d[6] = v - t[-2 * pitch + 2];  d[7] = v - t[-3 * pitch + 1];
    d[8] = v - t[-3 * pitch];         d[9] = v - t[-3 * pitch - 1];  d[10] = v - t[-2 * pitch - 2]; d[11] = v - t[-pitch - 3];
    d[12] = v - t[-3];                d[13] = v - t[pitch - 3];      d[14] = v - t[2 * pitch - 2];  d[15] = v - t[3 * pitch - 1];
    int mn2[16], mx2[16], mn4[16], mx4[16], mn8[16], mx8[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { mn8[k] = min(mn4[k], mn4[(k + 4) & 15]); mx8[k] = max(mx4[k], mx4[(k + 4) & 15]); }
    int A = -256, Bm = 256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        A = max(A, min(mn8[k], d[(k + 8) & 15]));        // darker arcs: min over 9 of (v - p)
        Bm = min(Bm, max(mx8[k], d[(k + 8) & 15]));      // brighter arcs: max over 9 of (v - p) -> -min(p - v)
    }
    return max(A, -Bm);
}

constexpr int kCellMax = 60;      // max cell edge: nCols = int(width/30) gives cells of 30..59 px (checked at configure time)

__global__ __launch_bounds__(256) void fast_cells_kernel(const DevGeom* __restrict__ G, const uint8_t* __restrict__ pyr,
                                                         int32_t* __restrict__ cell_cnt, uint32_t* __restrict__ cell_cand)
{
    __shared__ uint8_t tile[(kCellMax + 6) * (kCellMax + 8)];
    __shared__ uint8_t smap[(kCellMax + 2) * (kCellMax + 2)];
    __shared__ uint8_t s_cnt[16 * 4];        // keypoints per (pass of 256 pixels, wave)
    const int slice = blockIdx.y;
    // locate (level, cell)
    int cid = blockIdx.x, level = 0;
    while (level + 1 < G->nlevels && cid >= G->lv[level + 1].cell_off) level++;
    const LevelGeom& L = G->lv[level];
    const int local = cid - L.cell_off;
    const int ci = local / L.nCols, cj = local - ci * L.nCols;
    const int tid = threadIdx.x;
    int32_t* out_cnt = cell_cnt + (size_t)slice * G->ncells + cid;
    uint32_t* out = cell_cand + ((size_t)slice * G->ncells + cid) * G->cell_cap;
    // cell window (ORBextractor.cc:810-827); all quantities are exact small integers
    const int iniY = L.minBY + ci * L.hCell, iniX = L.minBX + cj * L.wCell;
    int maxY = iniY + L.hCell + 6, maxX = iniX + L.wCell + 6;
    if (iniY >= L.maxBY - 3 || iniX >= L.maxBX - 3) { if (tid == 0) *out_cnt = 0; return; }
    maxY = min(maxY, L.maxBY); maxX = min(maxX, L.maxBX);
    const int cw = maxX - iniX, ch = maxY - iniY;        // sub-image handed to cv::FAST
    const int dw = cw - 6, dh = ch - 6;                   // detection domain [3, cw-3) x [3, ch-3)
    if (dw <= 0 || dh <= 0) { if (tid == 0) *out_cnt = 0; return; }
    const uint8_t* roi = pyr + (size_t)slice * G->pyr_bytes + L.buf_off + (size_t)G->edge * L.bw + G->edge;
    const int pitch = kCellMax + 8;
    // A cell window of ONE grey value has no corner at any threshold >= 0 (every arc difference is 0): event images of short slices
    // are mostly such windows, and the arc scores (80 min / max per pixel) are the kernel's work.
    const uint8_t first = roi[(size_t)iniY * L.bw + iniX];
    int differs = 0;
    for (int i = tid; i < cw * ch; i += 256) {
        const int y = i / cw, x = i - y * cw;
        const uint8_t v = roi[(size_t)(iniY + y) * L.bw + iniX + x];
        tile[y * pitch + x] = v;
        differs |= (v != first);
    }
    if (!__syncthreads_or(differs) && G->iniTh >= 0 && G->minTh >= 0) { if (tid == 0) *out_cnt = 0; return; }
    // score plane with a zero ring: smap[(y+1)*sp + (x+1)] for domain pixel (x,y)
    const int sp = dw + 2;
    for (int i = tid; i < (dw + 2) * (dh + 2); i += 256) smap[i] = 0;
    __syncthreads();
    for (int i = tid; i < dw * dh; i += 256) {
        const int y = i / dw, x = i - y * dw;
        int S = fast_S(&tile[(y + 3) * pitch + x + 3], pitch);
        smap[(y + 1) * sp + x + 1] = (uint8_t)min(max(S, 0), 255);
    }
    __syncthreads();
    // keypoint test at threshold th: corner (S > th) and score S-1 strictly greater than the 8 neighbours'
    // scores, a non-corner or out-of-domain neighbour scoring 0 (cv::FAST buffers are zero initialised).
    int th = G->iniTh;
    const int npix = dw * dh, npass = (npix + 255) / 256;          // <= 15 passes of 256 pixels (cells of at most 60 x 60)
    const int w = tid >> 6, lane = tid & 63;
    for (int attempt = 0; attempt < 2; attempt++) {
        // raster order emission: every thread tests its pixel of every pass first (its flags stay in a register), the wave counts go
        // to LDS, and after ONE barrier every thread knows how many keypoints precede its pixel
        uint32_t flags = 0u;
        for (int p = 0; p < npass; p++) {
            const int i = p * 256 + tid;
            bool kp = false;
            if (i < npix) {
                const int y = i / dw, x = i - y * dw;
                const uint8_t* c = &smap[(y + 1) * sp + x + 1];
                const int S = c[0];
                if (S > th) {
                    const int sc = S - 1;
                    kp = true;
#pragma unroll
                    for (int oy = -1; oy <= 1; oy++)
#pragma unroll
                        for (int ox = -1; ox <= 1; ox++) {
                            if (ox == 0 && oy == 0) continue;
                            const int Sn = c[oy * sp + ox];
                            const int scn = (Sn > th) ? Sn - 1 : 0;
                            kp = kp && (sc > scn);
                        }
                }
            }
            const uint64_t bal = __ballot(kp);
            if (kp) flags |= 1u << p;
            if (lane == 0) s_cnt[p * 4 + w] = (uint8_t)__popcll(bal);
        }
        __syncthreads();
        int run = 0;
        for (int p = 0; p < npass; p++) {
            const int c0 = s_cnt[p * 4 + 0], c1 = s_cnt[p * 4 + 1], c2 = s_cnt[p * 4 + 2], c3 = s_cnt[p * 4 + 3];
            const bool kp = (flags >> p) & 1u;
            const uint64_t bal = __ballot(kp);
            if (kp) {
                const int base = run + (w > 0 ? c0 : 0) + (w > 1 ? c1 : 0) + (w > 2 ? c2 : 0);
                const int pos = base + __popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                const int i = p * 256 + tid, y = i / dw, x = i - y * dw;
                const int S = smap[(y + 1) * sp + x + 1];
                // keypoint coordinates relative to minBorder (:871-872): (x+3 + j*wCell, y+3 + i*hCell)
                const uint32_t kx = (uint32_t)(x + 3 + cj * L.wCell), ky = (uint32_t)(y + 3 + ci * L.hCell);
                if (pos < G->cell_cap) out[pos] = kx | (ky << 12) | ((uint32_t)(S - 1) << 24);
            }
            run += c0 + c1 + c2 + c3;
        }
        if (run > 0 || th == G->minTh) { if (tid == 0) *out_cnt = min(run, G->cell_cap); break; }
        th = G->minTh;        // vKeysCell.empty() -> retry with minThFAST (:849-852)
        if (attempt == 1 && tid == 0) *out_cnt = 0;
        __syncthreads();      // s_cnt is rewritten by the second attempt
    }
}

// ---------------------------------------------------------------------------------------------------
// octree: ORBextractor::DistributeOctTree (:558-782) in data-parallel rounds.
//
// The reference walks a std::list of nodes: a full pass divides every expandable node in list order (children pushed to the FRONT,
// parent erased); once "size + 3 * nToExpand > N" it divides the recorded (size, node) pairs largest first and stops the moment the
// list reaches N nodes.  All divisions of one pass are independent of each other -- only the order of the resulting list, the
// creation order of the children (tie-break of the size sort) and the stopping point depend on the processing order.  One round =
//   1. the nodes to divide, in processing order (a compaction of the expandable nodes / the sorted size list);
//   2. one wavefront per node: stable 4-way partition of its keys (DivideNode :500-556) into the other key buffer, child counts;
//   3. prefix sums over the processing order: where the pass stops (largest-first phase), creation index of every child;
//   4. the new list, written as an array: children in reverse creation order (what repeated push_front produces), then the
//      surviving nodes in their old order.
// The list IS the node array (node id = list position), rebuilt every round in a second array.  One workgroup per (slice, level).
struct ONode { uint16_t x0, x1, y0, y1, start, cnt, seq; uint8_t flags, pad; };     // flags: bit0 bNoMore, bit1 key buffer
static_assert(sizeof(ONode) == 16, "ONode is 16 bytes");
constexpr int kOctThreads = 512;
constexpr int kOctBigNode = 1024;   // keys from which a node is partitioned by the whole workgroup
constexpr int kOctBigMax = 64;      // (ncap <= 65 534 keys: at most 63 such nodes at a time)

__device__ __forceinline__ int oct_wave_incl_scan(int x)
{   // row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast 15 / 31
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

// exclusive prefix sums of a[0..n) in place (block-wide, any n); returns the total.  ws: kOctThreads / 64 + 2 ints of LDS
__device__ int oct_block_scan(int* a, int n, int* ws)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    int carry = 0;
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + tid;
        const int v = i < n ? a[i] : 0;
        const int incl = oct_wave_incl_scan(v);
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; w++) before += ws[w];
        int tot = 0;
        for (int w = 0; w < nw; w++) tot += ws[w];
        if (i < n) a[i] = before + incl - v;
        carry += tot;
        __syncthreads();
    }
    return carry;
}

// two exclusive prefix sums at once (the same barriers): a[0..na) and the 16-bit b[0..nb); returns a's total
__device__ int oct_block_scan2(int* a, int na, uint16_t* b, int nb, int* ws, int* ws2)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    int carry = 0, carryb = 0;
    const int n = max(na, nb);
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + tid;
        const int v = i < na ? a[i] : 0, vb = i < nb ? (int)b[i] : 0;
        const int incl = oct_wave_incl_scan(v), inclb = oct_wave_incl_scan(vb);
        if (lane == 63) { ws[wave] = incl; ws2[wave] = inclb; }
        __syncthreads();
        int before = carry, beforeb = carryb, tot = 0, totb = 0;
        for (int w = 0; w < nw; w++) { const int x = ws[w], y = ws2[w]; if (w < wave) { before += x; beforeb += y; } tot += x; totb += y; }
        if (i < na) a[i] = before + incl - v;
        if (i < nb) b[i] = (uint16_t)(beforeb + inclb - vb);
        carry += tot; carryb += totb;
        __syncthreads();
    }
    return carry;
}

// bitonic sort (ascending) of n u64 items by the whole workgroup; padded with ~0 up to the next power of two.  Thread t keeps
// items t, t + 512, ... in registers: partners less than 64 apart are exchanged by lane shuffles, partners a multiple of 512 apart
// are the thread's own registers, and only the distances 64, 128, 256 go through the array and two barriers (6 of the 45 steps of a
// 512-item sort).
__device__ __forceinline__ uint64_t oct_shfl_xor_u64(uint64_t v, int j)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, j, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), j, 64);
    return ((uint64_t)hi << 32) | lo;
}
template <int M>
__device__ void oct_block_sort_regs(uint64_t* a, int np2)
{
    const int tid = threadIdx.x;
    uint64_t v[M];
#pragma unroll
    for (int m = 0; m < M; m++) { const int i = tid + kOctThreads * m; v[m] = i < np2 ? a[i] : ~0ull; }
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= kOctThreads) {
#pragma unroll
                for (int dm = 1; dm < M; dm <<= 1) {                 // (static register indices)
                    if (j != dm * kOctThreads) continue;
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        if ((m & dm) == 0 && (m | dm) < M) {
                            const int i = tid + kOctThreads * m;
                            const bool up = (i & k) == 0;
                            const uint64_t x = v[m], y = v[m | dm];
                            const bool sw = (x > y) == up;
                            v[m] = sw ? y : x; v[m | dm] = sw ? x : y;
                        }
                    }
                }
            } else if (j >= 64) {
#pragma unroll
                for (int m = 0; m < M; m++) { const int i = tid + kOctThreads * m; if (i < np2) a[i] = v[m]; }
                __syncthreads();
                uint64_t o[M];
#pragma unroll
                for (int m = 0; m < M; m++) { const int i = tid + kOctThreads * m; o[m] = i < np2 ? a[i ^ j] : ~0ull; }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const int i = tid + kOctThreads * m;
                    const bool up = (i & k) == 0, low = (i & j) == 0;
                    const uint64_t mn = v[m] < o[m] ? v[m] : o[m], mx = v[m] < o[m] ? o[m] : v[m];
                    v[m] = (low == up) ? mn : mx;
                }
            } else {
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const int i = tid + kOctThreads * m;
                    const uint64_t o = oct_shfl_xor_u64(v[m], j);
                    const bool up = (i & k) == 0, low = (i & j) == 0;
                    const uint64_t mn = v[m] < o ? v[m] : o, mx = v[m] < o ? o : v[m];
                    v[m] = (low == up) ? mn : mx;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < M; m++) { const int i = tid + kOctThreads * m; if (i < np2) a[i] = v[m]; }
    __syncthreads();
}
// Up to one item per thread: sorted by RANK -- every thread counts the items below its own (all lanes read the same LDS word: a
// broadcast, no conflicts) and writes its item to that place.  n reads and compares per thread, no exchange network: 45 dependent
// shuffle / LDS steps of the bitonic sort against one pass.  The items must be distinct.  SH32: the items differ in bits [SH32 + 31 :
// SH32] already (cell paths; (size, creation order)), which makes the compare a 32-bit one; SH32 < 0: the whole 64 bits.
template <int SH32>
__device__ void oct_rank_sort_u64(uint64_t* a, int n)
{
    const int tid = threadIdx.x;
    const uint64_t v = tid < n ? a[tid] : ~0ull;
    int rank = 0;
    if (SH32 >= 0) {
        // the 32-bit keys packed at the front of the array (the items wait in registers): four keys per broadcast read
        const uint32_t mine = (uint32_t)(v >> SH32);
        const int n4 = (n + 3) & ~3;
        __syncthreads();
        if (tid < n4) ((uint32_t*)a)[tid] = tid < n ? mine : 0xffffffffu;     // (padding: never below anything)
        __syncthreads();
        // (only the wavefronts that hold an element count: the eight of the workgroup share one CU's four SIMDs, and the idle ones of a
        // 230-element sort doubled its time by walking the loop for nothing.  Four wavefronts for the whole kernel -- EORB_OCT_THREADS=256
        // in an experiment build -- lost 15 us per frame instead: the direct passes keep eight items per thread in registers)
        const uint4* k4 = (const uint4*)a;
        if (tid < n) {
#pragma unroll 4
            for (int j = 0; j < (n4 >> 2); j++) {
                const uint4 q = k4[j];
                rank += (q.x < mine ? 1 : 0) + (q.y < mine ? 1 : 0) + (q.z < mine ? 1 : 0) + (q.w < mine ? 1 : 0);
            }
        }
    } else if (tid < n) {
        const ulonglong2* k2 = (const ulonglong2*)a;
        const int n2 = n >> 1;
#pragma unroll 4
        for (int j = 0; j < n2; j++) { const ulonglong2 q = k2[j]; rank += (q.x < v ? 1 : 0) + (q.y < v ? 1 : 0); }
        if (n & 1) rank += (a[n - 1] < v) ? 1 : 0;
    }
    __syncthreads();
    if (tid < n) a[rank] = v;
    __syncthreads();
}
template <int SH32 = -1>
__device__ void oct_block_sort_u64(uint64_t* a, int n)
{
    if (n <= kOctThreads) { oct_rank_sort_u64<SH32>(a, n); return; }
    int np2 = 1; while (np2 < n) np2 <<= 1;
    for (int i = n + threadIdx.x; i < np2; i += blockDim.x) a[i] = ~0ull;
    __syncthreads();
    if (np2 <= kOctThreads) { oct_block_sort_regs<1>(a, np2); return; }
    if (np2 <= 2 * kOctThreads) { oct_block_sort_regs<2>(a, np2); return; }
    if (np2 <= 4 * kOctThreads) { oct_block_sort_regs<4>(a, np2); return; }
    if (np2 <= 8 * kOctThreads) { oct_block_sort_regs<8>(a, np2); return; }
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const uint64_t ai = a[i], al = a[l];
                    const bool up = (i & k) == 0;
                    if ((ai > al) == up) { a[i] = al; a[l] = ai; }
                }
            }
            __syncthreads();
        }
    }
}

// The cell path of a candidate: root index (bits 31..24), then twelve quadrant digits n1..n4 = 0..3 (bits 23..0, first division in the
// top two).  The boxes of DivideNode (:494-556) depend on the geometry only -- halfX = ceil((x1 - x0) / 2) -- so the path is a function
// of the position; the arithmetic is the list algorithm's own (float ceil of the integer extents, the roots' (int)(hX * r)).
__device__ __forceinline__ uint32_t oct_path_code(int px, int py, int nIni, float hX, int height)
{
    int r = (int)((float)px / hX);
    r = min(max(r, 0), nIni - 1);
    int x0 = (int)(uint16_t)(int)(hX * (float)r), x1 = (int)(uint16_t)(int)(hX * (float)(r + 1)), y0 = 0, y1 = height;
    uint32_t code = (uint32_t)r << 24;
#pragma unroll
    for (int d = 0; d < 12; d++) {
        const int halfX = (int)ceilf((float)(x1 - x0) / 2), halfY = (int)ceilf((float)(y1 - y0) / 2);
        const int midx = x0 + halfX, midy = y0 + halfY;
        const int q = (px < midx ? 0 : 1) + (py < midy ? 0 : 2);
        code |= (uint32_t)q << (22 - 2 * d);
        if (q & 1) x0 = midx; else x1 = midx;
        if (q & 2) y0 = midy; else y1 = midy;
    }
    return code;
}
// leading path levels two codes share: 0 = different roots, 1 = the root only, ..., 13 = all of it
__device__ __forceinline__ int oct_share(uint32_t a, uint32_t b)
{
    const uint32_t x = a ^ b;
    if (x >> 24) return 0;
    if (x == 0u) return 13;
    return 1 + ((__clz((int)x) - 8) >> 1);
}

// LDS_ONLY: every item of the working set sits in LDS (single frames, small levels).  The pointers are then known to be LDS
// pointers and the compiler emits ds_read / ds_write; with the mixed placement they are generic and every access is a FLAT
// instruction, whose round trip to the LDS is several times longer -- and a round is a chain of ~20 dependent accesses.
template <bool LDS_ONLY>
__global__ __launch_bounds__(kOctThreads) void octree_kernel(const DevGeom* __restrict__ G, const int32_t* __restrict__ cell_cnt,
                                                             const uint32_t* __restrict__ cell_cand, unsigned char* __restrict__ scratch_g,
                                                             uint32_t* __restrict__ lvl_kp, int32_t* __restrict__ lvl_cnt,
                                                             int32_t* __restrict__ err_flag, int32_t* __restrict__ sticky, int placement,
                                                             int32_t* __restrict__ redo /* LDS_ONLY with the dynamic placement: out, 1 = this level did not fit; else: in, only those levels run (null: all) */)
{
    extern __shared__ unsigned char smem[];
#ifdef EORB_OCT_TIMING      // (experiment builds only: where a call's cycles go, printed by workgroup 0)
    __shared__ long long s_tm[24]; __shared__ int s_tn[24];
    long long t_last = clock64();
    if (threadIdx.x < 24) { s_tm[threadIdx.x] = 0; s_tn[threadIdx.x] = 0; }
#define OCT_T(k) do { if (threadIdx.x == 0) { const long long t_now = clock64(); s_tm[k] += t_now - t_last; s_tn[k]++; t_last = t_now; } } while (0)
#else
#define OCT_T(k) do { } while (0)
#endif
    __shared__ int s_ws[kOctThreads / 64 + 2], s_ws2[kOctThreads / 64];
    __shared__ int s_m, s_flag, s_nbig;
    __shared__ uint16_t s_big[kOctBigMax];
    __shared__ int s_wc[kOctThreads / 64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = kOctThreads / 64;
    const int slice = blockIdx.x / G->nlevels, level = blockIdx.x % G->nlevels;
    const LevelGeom& L = G->lv[level];
    const int ncap = G->ncap_max, pool = L.node_cap, vcap = G->vsp_cap_max, pc = G->node_cap_max;
    // every item in LDS when the 160 KB hold it, otherwise in this (slice, level)'s block of the global scratch buffer: the
    // workgroup sits on one CU, whose own stores are visible to its later loads after __syncthreads()
    unsigned char* gblk = scratch_g + (size_t)blockIdx.x * G->oct_gblock_bytes[placement];
    if (!LDS_ONLY && redo && !redo[blockIdx.x]) return;             // (the LDS kernel did this level)
    // dynamic placement: the level's candidates are counted first, the key buffers and the points sized by them
    const bool dynp = LDS_ONLY && G->oct_dyn[placement];
    int o3 = 0, o4 = 0, o5 = 0;
    if (dynp) {
        const int nc0 = L.nCols * L.nRows;
        const int32_t* cc0 = cell_cnt + (size_t)slice * G->ncells + L.cell_off;
        int part = 0;
        for (int i = tid; i < nc0; i += kOctThreads) part += cc0[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        if (lane == 0) s_ws[wave] = part;
        __syncthreads();
        int ntot = 0;
        for (int w = 0; w < nwaves; w++) ntot += s_ws[w];
        __syncthreads();
        const int ncl = min(ntot, ncap), kb = (2 * ncl + 15) & ~15;
        o3 = G->oct_dyn_base[placement]; o4 = o3 + kb; o5 = o4 + kb;
        const bool fits = o5 + 4 * ncl <= G->oct_dyn_lds[placement];
        if (tid == 0 && redo) redo[blockIdx.x] = fits ? 0 : 1;
        if (!fits) return;
    }
    // (byte offsets, not pointers, are what is kept and selected between: under LDS_ONLY every pointer is then visibly smem + offset)
    auto item = [&](int k, int extra) -> unsigned char* {
        if (LDS_ONLY) {
            const int base = !dynp ? G->oct_off[placement][k] : (k == 3 ? o3 : (k == 4 ? o4 : (k == 5 ? o5 : G->oct_off_dyn[placement][k])));
            return smem + (base + extra);
        }
        return (G->oct_in_lds[placement][k] ? smem : gblk) + (G->oct_off[placement][k] + extra);
    };
    auto NODES = [&](int w) -> ONode* { return (ONode*)item(0, w * pc * (int)sizeof(ONode)); };
    auto VSP = [&](int w) -> uint64_t* { return (uint64_t*)item(1 + w, 0); };
    auto KEYS = [&](int w) -> uint16_t* { return (uint16_t*)item(3 + w, 0); };
    uint32_t* pts = (uint32_t*)item(5, 0);
    int* aux = (int*)item(6, 0);                    // pc ints: scan array (ne | n2 << 16 per processed node / survivor flags)
    auto PB = [&](int w) -> uint16_t* { return (uint16_t*)item(6, (4 + 2 * w) * pc); };      // 2 x pc: nodes to divide, in processing order (this pass | the next one)
    uint16_t* sv = (uint16_t*)item(6, 8 * pc);      // pc: full passes: the undivided nodes' ranks
    uint8_t* dv = (uint8_t*)item(6, 10 * pc);       // pc: cut-off passes: node is divided in this round
    uint64_t* cc = (uint64_t*)item(6, (11 * pc + 7) & ~7);      // pc: child counts of the processed nodes, 4 x 16 bits (item offsets are multiples of 16)
    int pb = 0, cut_stamp = 0;
    if (tid == 0) s_nbig = 0;
    for (int i = tid; i < pc; i += kOctThreads) dv[i] = 0;              // (the cut-off rounds' stamps; barriers follow before any use)
    uint32_t* lkp = lvl_kp + (size_t)slice * G->kp_total + L.kp_off;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    auto raise = [&](int bit) { if (tid == 0) { atomicOr(err_flag, bit); if (sticky) atomicOr(sticky, bit); } };

    // ---- gather the level's candidates in cell order (vToDistributeKeys) ----
    int n = 0;
    {
        const int nc = L.nCols * L.nRows;
        const int32_t* ccnt_g = cell_cnt + (size_t)slice * G->ncells + L.cell_off;
        if (nc <= pc) {
            // all cell counts at once (one load per thread, not one dependent load per cell), their offsets by a block scan, then a
            // wave per cell copies its candidates
            for (int i = tid; i < nc; i += kOctThreads) aux[i] = ccnt_g[i];
            __syncthreads();
            n = oct_block_scan(aux, nc, s_ws);
            for (int cidx = wave; cidx < nc; cidx += nwaves) {
                const int o = aux[cidx], ccnt = (cidx + 1 < nc ? aux[cidx + 1] : n) - o;
                const uint32_t* src = cell_cand + ((size_t)slice * G->ncells + L.cell_off + cidx) * G->cell_cap;
                for (int i = lane; i < ccnt; i += 64)
                    if (o + i < ncap) { pts[o + i] = src[i]; KEYS(0)[o + i] = (uint16_t)(o + i); }
            }
        } else {
            for (int cidx = 0; cidx < nc; cidx++) {
                const int ccnt = ccnt_g[cidx];
                const uint32_t* src = cell_cand + ((size_t)slice * G->ncells + L.cell_off + cidx) * G->cell_cap;
                for (int i = tid; i < ccnt; i += kOctThreads)
                    if (n + i < ncap) { pts[n + i] = src[i]; KEYS(0)[n + i] = (uint16_t)(n + i); }
                n += ccnt;
            }
        }
        if (n > ncap) { raise(1); n = ncap; }
    }
    __syncthreads();
    if (n == 0) { if (tid == 0) lvl_cnt[slice * G->nlevels + level] = 0; return; }
    OCT_T(0);

    const int N = L.nfeat;
    const int width = L.maxBX - L.minBX, height = L.maxBY - L.minBY;
    const int nIni = (int)roundf((float)width / (float)height);
    const float hX = (float)width / (float)nIni;
    int cur = 0, lsize = 0, seqctr = 0, nPnext = 0;
    bool overflow = false;
    int nvsp = 0, vw = 0;
    // ---- The full passes, computed instead of executed.  While no pass is cut short, EVERY node with more than one key is divided,
    // so after p passes a key sits in the cell of depth min(p, s) of its path, s = the depth at which it is alone.  Sorting the
    // path codes puts the keys of a cell side by side: the levels a key shares with its neighbours give s, the list size and the
    // number of divisible nodes after every pass (hence the pass that ends the loop or starts the cut-off stage :688-757) come from
    // two 13-bin histograms.  The LIST ORDER follows from push_front: after pass p the list is the pass's children in reverse
    // creation order, then the undivided (single-key) nodes in the order they had; creation order = the parents' list order, n1..n4
    // -- unrolled: a lexicographic order of the path digits whose directions alternate with the depth (below).  A second sort by
    // (that order, original key index) leaves the keys grouped by node, in list order, each node's keys in vToDistributeKeys order,
    // which is all the cut-off stage and the selection need.  A pass of the list algorithm is ~20 dependent LDS round trips; this is
    // two sorts in registers.  Levels whose keys do not fit the two size lists (the sort's buffer) keep the list algorithm. ----
    bool direct_done = false; int direct_cut = 0;
    __shared__ int s_hist[3][16];
    if (LDS_ONLY && n <= (dynp ? G->oct_dyn[placement] : G->oct_direct_cap[placement]) && nIni <= 255) {
        uint64_t* S = VSP(0);                       // (VSP(1) follows VSP(0): oct_direct_cap is 0 otherwise)
        uint16_t* sh = KEYS(1);
        if (tid < 48) s_hist[tid >> 4][tid & 15] = 0;
        for (int i = tid; i < n; i += kOctThreads) {
            const uint32_t p = pts[i];
            S[i] = ((uint64_t)oct_path_code((int)(p & 0xfff), (int)((p >> 12) & 0xfff), nIni, hX, height) << 16) | (uint64_t)i;
        }
        __syncthreads();
        OCT_T(8);
        oct_block_sort_u64<16>(S, n);                // (the cell paths of distinct pixels differ)
        OCT_T(9);
        for (int i = tid; i < n; i += kOctThreads) sh[i] = (uint16_t)(i > 0 ? oct_share((uint32_t)(S[i] >> 16), (uint32_t)(S[i - 1] >> 16)) : 0);
        __syncthreads();
        for (int i = tid; i < n; i += kOctThreads) {
            const int a = sh[i], b = i > 0 ? (int)sh[i - 1] : 0, nx = i + 1 < n ? (int)sh[i + 1] : 0;
            if (i > 0) {
                atomicAdd(&s_hist[0][a], 1);                              // pairs by shared levels
                if (a > b) { atomicAdd(&s_hist[1][b + 1], 1); atomicAdd(&s_hist[1][min(a + 1, 15)], -1); }     // a run of pairs sharing >= d levels starts here, for b < d <= a
            }
            atomicAdd(&s_hist[2][min(max(a, nx), 15)], 1);                // keys by the depth at which they are alone
        }
        __syncthreads();
        OCT_T(10);
        // L(p) = nodes after p passes, X(p) = those with more than one key, created(p) = nodes pass p made
        int cntge[16], runs[16], alone_lt[16];
        { int acc = 0; for (int d = 15; d >= 0; d--) { acc += s_hist[0][d]; cntge[d] = acc; } }
        { int acc = 0; for (int d = 0; d < 16; d++) { acc += s_hist[1][d]; runs[d] = acc; } }
        { int acc = 0; for (int d = 0; d < 16; d++) { alone_lt[d] = acc; acc += s_hist[2][d]; } }
        bool ok = s_hist[0][13] == 0 && s_hist[0][14] == 0 && s_hist[0][15] == 0;      // (two keys on one pixel: not a FAST output)
        int Pf = 0, ls = n - cntge[1], nx = runs[1], seqbase = 0, created = ls;
        if (ls > pool || nx > vcap) ok = false;
        bool fin = false;
        while (ok && !fin && !direct_cut) {
            const int prev = ls;
            if (nx == 0 || Pf >= 12) { fin = true; break; }                // nothing left to divide: the size does not change (:690)
            Pf++;
            seqbase += created;
            ls = n - cntge[Pf + 1]; nx = runs[Pf + 1]; created = ls - alone_lt[Pf];
            if (ls > pool || nx > vcap) { ok = false; break; }
            if (ls >= N || ls == prev) fin = true;
            else if (ls + nx * 3 > N) direct_cut = 1;
        }
        OCT_T(11);
        if (ok) {
            // the list-order key of every key's node
            uint64_t item[8];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int i = tid + kOctThreads * m;
                item[m] = ~0ull;
                if (i < n) {
                    const uint32_t code = (uint32_t)(S[i] >> 16);
                    const int a = sh[i], nxs = i + 1 < n ? (int)sh[i + 1] : 0;
                    const int e = min(Pf, max(a, nxs));
                    const int r = (int)(code >> 24);
                    uint32_t rk, qk = 0u;
                    if (e == 0) rk = (uint32_t)r;                                   // the roots were push_back'ed: ascending
                    else {
                        rk = ((e - 1) & 1) ? (uint32_t)r : (uint32_t)(nIni - 1 - r);
                        for (int k = 1; k <= e; k++) {
                            const uint32_t dgt = (code >> (24 - 2 * k)) & 3u;
                            qk |= (((e - k) & 1) ? dgt : 3u - dgt) << (24 - 2 * k);
                        }
                    }
                    item[m] = ((((uint64_t)(Pf - e) << 32) | ((uint64_t)rk << 24) | (uint64_t)qk) << 16) | (S[i] & 0xffffull);
                }
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 8; m++) { const int i = tid + kOctThreads * m; if (i < n) S[i] = item[m]; }
            __syncthreads();
            OCT_T(12);
            oct_block_sort_u64(S, n);
            OCT_T(13);
            // node boundaries -> list positions
            for (int i = tid; i < n; i += kOctThreads) sh[i] = (uint16_t)((i > 0 && (S[i] >> 16) != (S[i - 1] >> 16)) ? 1 : 0);
            __syncthreads();
            oct_block_scan2(aux, 0, sh, n, s_ws, s_ws2);                  // (exclusive: sh[i] = boundaries before i; + the boundary at i itself below)
            uint16_t* nstart = PB(0);
            for (int i = tid; i < n; i += kOctThreads) {
                const bool bnd = i == 0 || (S[i] >> 16) != (S[i - 1] >> 16);
                const int pos = (int)sh[i] + ((i > 0 && bnd) ? 1 : 0);
                KEYS(0)[i] = (uint16_t)(S[i] & 0xffffull);
                if (bnd) nstart[pos] = (uint16_t)i;
            }
            __syncthreads();
            OCT_T(14);
            // the nodes, in list order; the divisible ones also go to the size list of the cut-off stage
            for (int pos = tid; pos < ls; pos += kOctThreads) aux[pos] = 0;
            __syncthreads();
            for (int pos = tid; pos < ls; pos += kOctThreads) {
                const int st = nstart[pos], cnt = (pos + 1 < ls ? (int)nstart[pos + 1] : n) - st;
                const uint64_t it = S[st];
                const int e = Pf - (int)((it >> 48) & 0xf);
                const uint32_t p = pts[it & 0xffffull];
                const int px = (int)(p & 0xfff), py = (int)((p >> 12) & 0xfff);
                int r = (int)((float)px / hX);
                r = min(max(r, 0), nIni - 1);
                int x0 = (int)(uint16_t)(int)(hX * (float)r), x1 = (int)(uint16_t)(int)(hX * (float)(r + 1)), y0 = 0, y1 = height;
                for (int d = 0; d < e; d++) {
                    const int halfX = (int)ceilf((float)(x1 - x0) / 2), halfY = (int)ceilf((float)(y1 - y0) / 2);
                    const int midx = x0 + halfX, midy = y0 + halfY;
                    if (px < midx) x1 = midx; else x0 = midx;
                    if (py < midy) y1 = midy; else y0 = midy;
                }
                ONode nd;
                nd.x0 = (uint16_t)x0; nd.x1 = (uint16_t)x1; nd.y0 = (uint16_t)y0; nd.y1 = (uint16_t)y1;
                nd.start = (uint16_t)st; nd.cnt = (uint16_t)cnt;
                const int seq = e == Pf ? (Pf == 0 ? pos : seqbase + created - 1 - pos) : 0;
                nd.seq = (uint16_t)seq; nd.flags = (uint8_t)(cnt == 1 ? 1 : 0); nd.pad = 0;
                NODES(0)[pos] = nd;
                if (cnt > 1) aux[pos] = 1;
            }
            __syncthreads();
            OCT_T(15);
            // (S = VSP(0..1) is free now: the size list is written over it)
            oct_block_scan(aux, ls, s_ws);
            for (int pos = tid; pos < ls; pos += kOctThreads) {
                const ONode nd = NODES(0)[pos];
                if (nd.cnt > 1) VSP(0)[aux[pos]] = ((uint64_t)nd.cnt << 32) | ((uint64_t)nd.seq << 16) | (uint64_t)pos;
            }
            __syncthreads();
            cur = 0; lsize = ls; seqctr = seqbase + created; nvsp = nx; vw = 0; pb = 0;
            direct_done = true;
        } else direct_cut = 0;
    }
    OCT_T(6);
    // ---- root nodes :562-603: stable partition of the keys by root index (wave 0), empty roots erased, singletons bNoMore ----
    if (!direct_done) {
        int off = 0;
        for (int r = 0; r < nIni; r++) {
            int c = 0;
            if (wave == 0) {
                for (int k0 = 0; k0 < n; k0 += 64) {
                    const int i = k0 + lane;
                    bool mine = false; uint16_t key = 0;
                    if (i < n) {
                        key = KEYS(0)[i];
                        const float kx = (float)(pts[key] & 0xfff);
                        int b = (int)(kx / hX);
                        b = min(max(b, 0), nIni - 1);
                        mine = (b == r);
                    }
                    const uint64_t bal = __ballot(mine);
                    if (mine) KEYS(1)[off + c + __popcll(bal & lt_mask)] = key;
                    c += __popcll(bal);
                }
                if (lane == 0) s_m = c;
            }
            __syncthreads();
            c = s_m;
            if (c > 0) {
                if (tid == 0) {
                    ONode nd;
                    nd.x0 = (uint16_t)(int)(hX * (float)r); nd.x1 = (uint16_t)(int)(hX * (float)(r + 1));
                    nd.y0 = 0; nd.y1 = (uint16_t)height;
                    nd.start = (uint16_t)off; nd.cnt = (uint16_t)c; nd.seq = (uint16_t)seqctr;
                    nd.flags = (uint8_t)((c == 1 ? 1 : 0) | 2); nd.pad = 0;
                    NODES(0)[lsize] = nd;                           // push_back
                    if (c > 1) PB(0)[nPnext] = (uint16_t)lsize;     // (the first pass walks the roots in list order)
                }
                if (c > 1) nPnext++;
                lsize++; seqctr++;
            }
            off += c;
            __syncthreads();
        }
    }

    OCT_T(1);
    // One round: divide the nodes P[0..nP) (processing order).  cut: stop after the division that brings the list to N nodes
    // (:741-757).  Returns through the references; VSP(vw) receives the children with more than one key, in creation order.
    auto round = [&](int nP, bool cut, int& nToExpand) {
        const ONode* src = NODES(cur);
        ONode* dst = NODES(cur ^ 1);
        const uint16_t* P = PB(pb);
        uint16_t* Pn = PB(pb ^ 1);
        // 2a. nodes with many keys (the first passes: one or two roots hold every candidate): the whole workgroup partitions one
        //     node, wave w its w-th contiguous share of the keys; the shares' class counts meet in LDS
        //     (no node can be that large on a level with fewer candidates: the two barriers are skipped, s_nbig stays 0)
        int nbig = 0;
        if (n >= kOctBigNode) {
            if (tid == 0) s_nbig = 0;
            __syncthreads();
            for (int j = tid; j < nP; j += kOctThreads)
                if (src[P[j]].cnt >= kOctBigNode) { const int k = atomicAdd(&s_nbig, 1); if (k < kOctBigMax) s_big[k] = (uint16_t)j; }
            __syncthreads();
            nbig = min(s_nbig, kOctBigMax);                         // (at most ncap / kOctBigNode <= 64 such nodes exist)
        }
        for (int b = 0; b < nbig; b++) {
            const int j = s_big[b];
            const ONode nd = src[P[j]];
            const int sb = (nd.flags >> 1) & 1;
            const uint16_t* ks = KEYS(sb); uint16_t* kd = KEYS(sb ^ 1);
            const int halfX = (int)ceilf((float)(nd.x1 - nd.x0) / 2), halfY = (int)ceilf((float)(nd.y1 - nd.y0) / 2);
            const int midx = nd.x0 + halfX, midy = nd.y0 + halfY;
            const int start = nd.start, cnt = nd.cnt;
            const int share = (((cnt + nwaves - 1) / nwaves) + 63) & ~63;          // keys per wave, a multiple of 64
            const int k_lo = min(wave * share, cnt), k_hi = min(k_lo + share, cnt);
            int c[4] = {0, 0, 0, 0};
            for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
                const int i = k0 + lane;
                int cls = -1;
                if (i < k_hi) {
                    const uint32_t p = pts[ks[start + i]];
                    const int px = p & 0xfff, py = (p >> 12) & 0xfff;
                    cls = (px < midx ? 0 : 1) + (py < midy ? 0 : 2);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) c[q] += __popcll(__ballot(cls == q));
            }
            if (lane == 0) { s_wc[wave][0] = c[0]; s_wc[wave][1] = c[1]; s_wc[wave][2] = c[2]; s_wc[wave][3] = c[3]; }
            __syncthreads();
            int tot[4] = {0, 0, 0, 0}, before[4] = {0, 0, 0, 0};
            for (int w = 0; w < nwaves; w++)
#pragma unroll
                for (int q = 0; q < 4; q++) { const int v = s_wc[w][q]; tot[q] += v; if (w < wave) before[q] += v; }
            int sbase[4]; sbase[0] = start; sbase[1] = sbase[0] + tot[0]; sbase[2] = sbase[1] + tot[1]; sbase[3] = sbase[2] + tot[2];
            int run[4] = {before[0], before[1], before[2], before[3]};
            for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
                const int i = k0 + lane;
                int cls = -1; uint16_t key = 0;
                if (i < k_hi) {
                    key = ks[start + i];
                    const uint32_t p = pts[key];
                    const int px = p & 0xfff, py = (p >> 12) & 0xfff;
                    cls = (px < midx ? 0 : 1) + (py < midy ? 0 : 2);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint64_t bal = __ballot(cls == q);
                    if (cls == q) kd[sbase[q] + run[q] + __popcll(bal & lt_mask)] = key;
                    run[q] += __popcll(bal);
                }
            }
            if (tid == 0) {
                cc[j] = (uint64_t)tot[0] | ((uint64_t)tot[1] << 16) | ((uint64_t)tot[2] << 32) | ((uint64_t)tot[3] << 48);
                aux[j] = ((tot[0] > 0) + (tot[1] > 0) + (tot[2] > 0) + (tot[3] > 0)) | (((tot[0] > 1) + (tot[1] > 1) + (tot[2] > 1) + (tot[3] > 1)) << 16);
            }
            __syncthreads();                                        // s_wc is rewritten by the next node
        }
        // 2b. the remaining nodes: child counts, stable partition into the other key buffer (DivideNode :531-545).  A wave takes four
        //     nodes at a time, one per 16-lane group, when they have at most 16 keys (after the first passes nearly all do: a wave per
        //     node would run a quarter full and walk the nodes one latency chain at a time); larger ones get the whole wave below.
        for (int j0 = wave * 4; j0 < nP; j0 += nwaves * 4) {
            const int grp = lane >> 4, gl = lane & 15;
            const int jg = j0 + grp;
            ONode nd; nd.cnt = 0; nd.flags = 0; nd.start = 0; nd.x0 = nd.x1 = nd.y0 = nd.y1 = 0;
            if (jg < nP) nd = src[P[jg]];
            const bool small = jg < nP && nd.cnt <= 16;
            {
                const int sb = (nd.flags >> 1) & 1;
                const int halfX = (int)ceilf((float)(nd.x1 - nd.x0) / 2), halfY = (int)ceilf((float)(nd.y1 - nd.y0) / 2);
                const int midx = nd.x0 + halfX, midy = nd.y0 + halfY;
                int cls = -1; uint16_t key = 0;
                if (small && gl < nd.cnt) {
                    key = KEYS(sb)[nd.start + gl];
                    const uint32_t p = pts[key];
                    cls = ((int)(p & 0xfff) < midx ? 0 : 1) + ((int)((p >> 12) & 0xfff) < midy ? 0 : 2);
                }
                const int sh = grp * 16;
                const uint32_t b0 = (uint32_t)(__ballot(cls == 0) >> sh) & 0xffffu, b1 = (uint32_t)(__ballot(cls == 1) >> sh) & 0xffffu,
                               b2 = (uint32_t)(__ballot(cls == 2) >> sh) & 0xffffu, b3 = (uint32_t)(__ballot(cls == 3) >> sh) & 0xffffu;
                const int c0 = __popc(b0), c1 = __popc(b1), c2 = __popc(b2), c3 = __popc(b3);
                if (cls >= 0) {
                    const uint32_t mb = cls == 0 ? b0 : (cls == 1 ? b1 : (cls == 2 ? b2 : b3));
                    const int base = nd.start + (cls > 0 ? c0 : 0) + (cls > 1 ? c1 : 0) + (cls > 2 ? c2 : 0);
                    KEYS(sb ^ 1)[base + __popc(mb & ((1u << gl) - 1u))] = key;
                }
                if (small && gl == 0) {
                    cc[jg] = (uint64_t)c0 | ((uint64_t)c1 << 16) | ((uint64_t)c2 << 32) | ((uint64_t)c3 << 48);
                    aux[jg] = ((c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0)) | (((c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1)) << 16);
                }
            }
            // the nodes of this quartet with more than 16 keys, one after the other with the whole wave
            const uint64_t bigm = __ballot(jg < nP && !small && gl == 0);
          for (int g = 0; g < 4; g++) {
            if (!((bigm >> (16 * g)) & 1ull)) continue;
            const int j = j0 + g;
            const ONode nd = src[P[j]];
            if (nd.cnt >= kOctBigNode && s_nbig <= kOctBigMax) continue;
            if (nd.cnt <= 64) {
                // (most nodes: one sub-batch -- the four ballots give the counts and the ranks at once)
                const int sb = (nd.flags >> 1) & 1;
                const int halfX = (int)ceilf((float)(nd.x1 - nd.x0) / 2), halfY = (int)ceilf((float)(nd.y1 - nd.y0) / 2);
                const int midx = nd.x0 + halfX, midy = nd.y0 + halfY;
                int cls = -1; uint16_t key = 0;
                if (lane < nd.cnt) {
                    key = KEYS(sb)[nd.start + lane];
                    const uint32_t p = pts[key];
                    cls = ((int)(p & 0xfff) < midx ? 0 : 1) + ((int)((p >> 12) & 0xfff) < midy ? 0 : 2);
                }
                const uint64_t b0 = __ballot(cls == 0), b1 = __ballot(cls == 1), b2 = __ballot(cls == 2), b3 = __ballot(cls == 3);
                const int c0 = __popcll(b0), c1 = __popcll(b1), c2 = __popcll(b2), c3 = __popcll(b3);
                if (cls >= 0) {
                    const uint64_t mb = cls == 0 ? b0 : (cls == 1 ? b1 : (cls == 2 ? b2 : b3));
                    const int base = nd.start + (cls > 0 ? c0 : 0) + (cls > 1 ? c1 : 0) + (cls > 2 ? c2 : 0);
                    KEYS(sb ^ 1)[base + __popcll(mb & lt_mask)] = key;
                }
                if (lane == 0) {
                    cc[j] = (uint64_t)c0 | ((uint64_t)c1 << 16) | ((uint64_t)c2 << 32) | ((uint64_t)c3 << 48);
                    aux[j] = ((c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0)) | (((c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1)) << 16);
                }
                continue;
            }
            const int sb = (nd.flags >> 1) & 1;
            const uint16_t* ks = KEYS(sb); uint16_t* kd = KEYS(sb ^ 1);
            const int halfX = (int)ceilf((float)(nd.x1 - nd.x0) / 2), halfY = (int)ceilf((float)(nd.y1 - nd.y0) / 2);
            const int midx = nd.x0 + halfX, midy = nd.y0 + halfY;
            const int start = nd.start, cnt = nd.cnt;
            int c[4] = {0, 0, 0, 0};
            for (int k0 = 0; k0 < cnt; k0 += 64) {
                const int i = k0 + lane;
                int cls = -1;
                if (i < cnt) {
                    const uint32_t p = pts[ks[start + i]];
                    const int px = p & 0xfff, py = (p >> 12) & 0xfff;
                    cls = (px < midx ? 0 : 1) + (py < midy ? 0 : 2);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) c[q] += __popcll(__ballot(cls == q));
            }
            int sbase[4]; sbase[0] = start; sbase[1] = sbase[0] + c[0]; sbase[2] = sbase[1] + c[1]; sbase[3] = sbase[2] + c[2];
            int run[4] = {0, 0, 0, 0};
            for (int k0 = 0; k0 < cnt; k0 += 64) {
                const int i = k0 + lane;
                int cls = -1; uint16_t key = 0;
                if (i < cnt) {
                    key = ks[start + i];
                    const uint32_t p = pts[key];
                    const int px = p & 0xfff, py = (p >> 12) & 0xfff;
                    cls = (px < midx ? 0 : 1) + (py < midy ? 0 : 2);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint64_t bal = __ballot(cls == q);
                    if (cls == q) kd[sbase[q] + run[q] + __popcll(bal & lt_mask)] = key;
                    run[q] += __popcll(bal);
                }
            }
            if (lane == 0) {
                cc[j] = (uint64_t)c[0] | ((uint64_t)c[1] << 16) | ((uint64_t)c[2] << 32) | ((uint64_t)c[3] << 48);
                aux[j] = ((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0)) | (((c[0] > 1) + (c[1] > 1) + (c[2] > 1) + (c[3] > 1)) << 16);
            }
          }
        }
        if (!cut) {
            // ---- a full pass divides every node of P = every node with more than one key: the survivors are the bNoMore nodes.  Both
            //      prefix sums (children over the processing order, survivors over the list) share their barriers, the next pass's
            //      processing order falls out of the children's placement: four barriers per pass ----
            for (int i = tid; i < lsize; i += kOctThreads) sv[i] = (uint16_t)(src[i].flags & 1);
            __syncthreads();
            const int tot = oct_block_scan2(aux, nP, sv, lsize, s_ws, s_ws2);
            const int C = tot & 0xffff, n2tot = tot >> 16, nsurv = lsize - nP;
            if (C + nsurv > pool || n2tot > vcap) { overflow = true; nToExpand = 0; return; }   // uniform: every thread sees the same numbers
            uint64_t* vout = VSP(vw);
            for (int j = tid; j < nP; j += kOctThreads) {
                const ONode nd = src[P[j]];
                const uint64_t cj = cc[j];
                const int c[4] = {(int)(cj & 0xffff), (int)((cj >> 16) & 0xffff), (int)((cj >> 32) & 0xffff), (int)(cj >> 48)};
                const int halfX = (int)ceilf((float)(nd.x1 - nd.x0) / 2), halfY = (int)ceilf((float)(nd.y1 - nd.y0) / 2);
                const int midx = nd.x0 + halfX, midy = nd.y0 + halfY;
                const int sb = (nd.flags >> 1) & 1;
                int k = aux[j] & 0xffff, k2 = aux[j] >> 16, sb0 = nd.start;
#pragma unroll
                for (int q = 0; q < 4; q++) {                       // n1..n4: x-low/y-low, x-high/y-low, x-low/y-high, x-high/y-high
                    if (c[q] > 0) {
                        ONode ch;
                        ch.x0 = (uint16_t)((q & 1) ? midx : nd.x0); ch.x1 = (uint16_t)((q & 1) ? nd.x1 : midx);
                        ch.y0 = (uint16_t)((q & 2) ? midy : nd.y0); ch.y1 = (uint16_t)((q & 2) ? nd.y1 : midy);
                        ch.start = (uint16_t)sb0; ch.cnt = (uint16_t)c[q]; ch.seq = (uint16_t)(seqctr + k);
                        ch.flags = (uint8_t)((c[q] == 1 ? 1 : 0) | ((sb ^ 1) << 1)); ch.pad = 0;
                        const int pos = C - 1 - k;                  // push_front in creation order
                        dst[pos] = ch;
                        if (c[q] > 1) {
                            vout[k2] = ((uint64_t)c[q] << 32) | ((uint64_t)(uint16_t)(seqctr + k) << 16) | (uint64_t)pos;
                            Pn[n2tot - 1 - k2] = (uint16_t)pos;     // list order of the next pass = the reverse of the creation order
                            k2++;
                        }
                        k++;
                    }
                    sb0 += c[q];
                }
            }
            for (int i = tid; i < lsize; i += kOctThreads) if (src[i].flags & 1) dst[C + sv[i]] = src[i];
            __syncthreads();
            seqctr += C; lsize = C + nsurv; cur ^= 1; nvsp = n2tot; nToExpand = n2tot; nPnext = n2tot; pb ^= 1;
            return;
        }
        if (tid == 0) s_m = nP;
        __syncthreads();
        // 3. prefix sums over the processing order: aux[j] = children (low half) / children with > 1 key (high half) before j
        const int tot = oct_block_scan(aux, nP, s_ws);
        if (cut) {
            // list size after processing j = lsize + (children of 0..j) - (j + 1); the pass stops after the first j reaching N
            for (int j = tid; j < nP; j += kOctThreads) {
                const uint64_t cj = cc[j];
                const int ne = ((cj & 0xffff) != 0) + (((cj >> 16) & 0xffff) != 0) + (((cj >> 32) & 0xffff) != 0) + ((cj >> 48) != 0);
                if (lsize + (aux[j] & 0xffff) + ne - (j + 1) >= N) atomicMin(&s_m, j + 1);
            }
            __syncthreads();
        }
        const int m = s_m;
        int C, n2tot;
        if (m == nP) { C = tot & 0xffff; n2tot = tot >> 16; }
        else { C = aux[m] & 0xffff; n2tot = aux[m] >> 16; }
        // 4a. survivors: old nodes that are not divided keep their relative order behind the children.  dv[i] == this round's stamp marks
        //     a divided node (no clearing between rounds; the array is cleared when the 8-bit stamp wraps)
        if (++cut_stamp == 256) {
            for (int i = tid; i < pc; i += kOctThreads) dv[i] = 0;
            cut_stamp = 1;
            __syncthreads();
        }
        const uint8_t stamp = (uint8_t)cut_stamp;
        const int nsurv = lsize - m;
        if (C + nsurv > pool || n2tot > vcap) { overflow = true; nToExpand = 0; return; }   // uniform: every thread sees the same numbers
        for (int j = tid; j < m; j += kOctThreads) dv[P[j]] = stamp;
        // the children in the same phase (they only need aux / cc / src)
        uint64_t* vout = VSP(vw);
        for (int j = tid; j < m; j += kOctThreads) {
            const ONode nd = src[P[j]];
            const uint64_t cj = cc[j];
            const int c[4] = {(int)(cj & 0xffff), (int)((cj >> 16) & 0xffff), (int)((cj >> 32) & 0xffff), (int)(cj >> 48)};
            const int halfX = (int)ceilf((float)(nd.x1 - nd.x0) / 2), halfY = (int)ceilf((float)(nd.y1 - nd.y0) / 2);
            const int midx = nd.x0 + halfX, midy = nd.y0 + halfY;
            const int sb = (nd.flags >> 1) & 1;
            int k = aux[j] & 0xffff, k2 = aux[j] >> 16, sb0 = nd.start;
#pragma unroll
            for (int q = 0; q < 4; q++) {                           // n1..n4: x-low/y-low, x-high/y-low, x-low/y-high, x-high/y-high
                if (c[q] > 0) {
                    ONode ch;
                    ch.x0 = (uint16_t)((q & 1) ? midx : nd.x0); ch.x1 = (uint16_t)((q & 1) ? nd.x1 : midx);
                    ch.y0 = (uint16_t)((q & 2) ? midy : nd.y0); ch.y1 = (uint16_t)((q & 2) ? nd.y1 : midy);
                    ch.start = (uint16_t)sb0; ch.cnt = (uint16_t)c[q]; ch.seq = (uint16_t)(seqctr + k);
                    ch.flags = (uint8_t)((c[q] == 1 ? 1 : 0) | ((sb ^ 1) << 1)); ch.pad = 0;
                    const int pos = C - 1 - k;                      // push_front in creation order
                    dst[pos] = ch;
                    if (c[q] > 1) { vout[k2] = ((uint64_t)c[q] << 32) | ((uint64_t)(uint16_t)(seqctr + k) << 16) | (uint64_t)pos; k2++; }
                    k++;
                }
                sb0 += c[q];
            }
        }
        __syncthreads();
        for (int i = tid; i < lsize; i += kOctThreads) sv[i] = dv[i] == stamp ? 0 : 1;       // (read back by the same thread in the scan)
        oct_block_scan2(aux, 0, sv, lsize, s_ws, s_ws2);
        for (int i = tid; i < lsize; i += kOctThreads) if (dv[i] != stamp) dst[C + sv[i]] = src[i];
        __syncthreads();
        seqctr += C; lsize = C + nsurv; cur ^= 1; nvsp = n2tot; nToExpand = n2tot;
    };

    bool finish = false;
    // the cut-off stage :688-757: the divisible nodes largest first (ascending sort of (size, creation order), walked from the back
    // :703-712), one at a time until the list holds N nodes; repeated on the children while that changes anything
    auto cut_stage = [&]() {
        int guard2 = 0;
        while (!finish && !overflow && guard2++ < 4096) {
            const int prevSize2 = lsize;
            const int nprev = nvsp;
            oct_block_sort_u64<16>(VSP(vw), nprev);     // (size << 16 | creation order: distinct, 32 bits)
            OCT_T(3);
            for (int j = tid; j < nprev; j += kOctThreads) PB(pb)[j] = (uint16_t)(VSP(vw)[nprev - 1 - j] & 0xffff);
            vw ^= 1;
            __syncthreads();
            int dummy = 0;
            if (nprev > 0) round(nprev, true, dummy);
            OCT_T(4);
            if (lsize >= N || lsize == prevSize2) finish = true;
        }
    };
    if (direct_done) {
        if (direct_cut) cut_stage();
    } else {
        int guard = 0;
        while (!finish && !overflow && guard++ < 64) {
            const int prevSize = lsize;
            // processing order of a full pass = list order of the expandable nodes (:618-686): left in PB(pb) by the roots / the previous pass
            const int nP = nPnext;
            int nToExpand = 0;
            if (nP > 0) round(nP, false, nToExpand);
            OCT_T(2);
            if (overflow) break;
            if (lsize >= N || lsize == prevSize) finish = true;
            else if (lsize + nToExpand * 3 > N) cut_stage();
        }
    }
    if (overflow) raise(2);

    // ---- retain the best point of each node, in list order (:760-780) ----
    {
        const ONode* nd = NODES(cur);
        const int nout = min(lsize, L.kp_cap);
        if (lsize > L.kp_cap) raise(4);
        for (int i = tid; i < nout; i += kOctThreads) {
            const uint16_t* ks = KEYS((nd[i].flags >> 1) & 1) + nd[i].start;
            const int cnt = nd[i].cnt;
            uint32_t best = pts[ks[0]];
            for (int q = 1; q < cnt; q++) {
                const uint32_t p = pts[ks[q]];
                if ((p >> 24) > (best >> 24)) best = p;
            }
            lkp[i] = best;
        }
        if (tid == 0) lvl_cnt[slice * G->nlevels + level] = nout;
    }
    OCT_T(5);
#ifdef EORB_OCT_TIMING
    if (threadIdx.x == 0 && slice == 0)
        printf("   direct: codes %lld sort1 %lld shares+hist %lld sim %lld items %lld sort2 %lld bounds %lld nodes %lld vsp %lld\n", s_tm[8], s_tm[9], s_tm[10], s_tm[11], s_tm[12], s_tm[13], s_tm[14], s_tm[15], s_tm[6]);
    if (threadIdx.x == 0 && slice == 0)
        printf("octree L%d n=%d N=%d lsize=%d direct %d (cap %d) %lld: gather %lld | roots %lld | full passes %d x %lld | cut sorts %d x %lld | cut rounds %d x %lld | select %lld cycles\n", level, n, N, lsize, (int)direct_done + 10 * (int)dynp + 100 * (int)LDS_ONLY, G->oct_direct_cap[placement], s_tm[6],
               s_tm[0], s_tm[1], s_tn[2], s_tn[2] ? s_tm[2] / s_tn[2] : 0, s_tn[3], s_tn[3] ? s_tm[3] / s_tn[3] : 0, s_tn[4], s_tn[4] ? s_tm[4] / s_tn[4] : 0, s_tm[5]);
#endif
}

// ---------------------------------------------------------------------------------------------------
// orientation: IC_Angle :77-104.  32 lanes per keypoint: lane = column u of the disc (-15 .. 15, one lane idle), walking the row pairs
// +-v: consecutive lanes read consecutive bytes of a row (with a lane per ROW every load instruction of a wave touched 64 different
// image lines).  Integer moments: the order of the sums is free.
__global__ __launch_bounds__(256) void orient_kernel(const DevGeom* __restrict__ G, const uint8_t* __restrict__ pyr,
                                                     const uint32_t* __restrict__ lvl_kp, const int32_t* __restrict__ lvl_cnt,
                                                     float* __restrict__ kp_angle)
{
    const int slice = blockIdx.y;
    const int gid = blockIdx.x * 8 + (threadIdx.x >> 5);       // keypoint slot in the slice's level arrays
    const int u = (int)(threadIdx.x & 31) - 15;                // -15 .. 16 (16: no column)
    if (gid >= G->kp_total) return;
    int level = 0;
    while (level + 1 < G->nlevels && gid >= G->lv[level + 1].kp_off) level++;
    const LevelGeom& L = G->lv[level];
    const int i = gid - L.kp_off;
    if (i >= lvl_cnt[slice * G->nlevels + level]) return;     // whole 32-lane group exits together
    const uint32_t p = lvl_kp[(size_t)slice * G->kp_total + gid];
    // keypoint in level coordinates: candidate coords are relative to minBorder (:889-893)
    const int cx = (int)(p & 0xfff) + L.minBX, cy = (int)((p >> 12) & 0xfff) + L.minBY;
    const uint8_t* center = pyr + (size_t)slice * G->pyr_bytes + L.buf_off + (size_t)(cy + G->edge) * L.bw + (cx + G->edge);
    const int step = L.bw;
    const int au = u < 0 ? -u : u;
    int m_01 = 0, m_10 = 0;
    if (au <= 15) m_10 = u * (int)center[u];
#pragma unroll
    for (int v = 1; v <= 15; ++v) {
        if (au <= G->umax[v]) {
            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
            m_01 += v * (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {
        m_01 += __shfl_xor(m_01, d, 64);
        m_10 += __shfl_xor(m_10, d, 64);
    }
    if ((threadIdx.x & 31) == 0) kp_angle[(size_t)slice * G->kp_total + gid] = dev_fast_atan2((float)m_01, (float)m_10);
}

// ---------------------------------------------------------------------------------------------------
// GaussianBlur 5x5 sigma 2 (Q8 coefficients 39 57 64 57 39), REFLECT_101, on the un-bordered level
constexpr int kBlurTW = 64, kBlurTH = 16;        // outputs per workgroup (32 x 8 tiles: 74 000 workgroups per 128 frames, the launch bound by their dispatch)
__global__ __launch_bounds__(256) void blur_kernel(const DevGeom* __restrict__ G, const uint8_t* __restrict__ pyr,
                                                   uint8_t* __restrict__ blur)
{
    __shared__ int rows[(kBlurTH + 4) * kBlurTW];
    const int slice = blockIdx.z;
    // blockIdx.y encodes (level, tile row); blockIdx.x tile column
    int level = 0, ty = blockIdx.y;
    while (level < G->nlevels) {
        const int nty = (G->lv[level].h + kBlurTH - 1) / kBlurTH;
        if (ty < nty) break;
        ty -= nty; level++;
    }
    if (level >= G->nlevels) return;
    const LevelGeom& L = G->lv[level];
    const int x0 = blockIdx.x * kBlurTW, y0 = ty * kBlurTH;
    if (x0 >= L.w) return;
    const uint8_t* roi = pyr + (size_t)slice * G->pyr_bytes + L.buf_off + (size_t)G->edge * L.bw + G->edge;
    const int k0 = 39, k1 = 57, k2 = 64;
    // horizontal pass for TH + 4 source rows x TW columns
    for (int i = threadIdx.x; i < (kBlurTH + 4) * kBlurTW; i += 256) {
        const int r = i / kBlurTW, cxi = i - r * kBlurTW;
        const int sy = reflect101(y0 + r - 2, L.h);
        const int x = x0 + cxi;
        int acc = 0;
        if (x < L.w) {
            const uint8_t* Srow = roi + (size_t)sy * L.bw;
            acc = k0 * Srow[reflect101(x - 2, L.w)] + k1 * Srow[reflect101(x - 1, L.w)] + k2 * Srow[x] +
                  k1 * Srow[reflect101(x + 1, L.w)] + k0 * Srow[reflect101(x + 2, L.w)];
        }
        rows[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBlurTH * kBlurTW; i += 256) {
        const int ly = i / kBlurTW, lx = i - ly * kBlurTW;
        const int x = x0 + lx, y = y0 + ly;
        if (x < L.w && y < L.h) {
            const int acc = k0 * rows[(ly + 0) * kBlurTW + lx] + k1 * rows[(ly + 1) * kBlurTW + lx] + k2 * rows[(ly + 2) * kBlurTW + lx] +
                            k1 * rows[(ly + 3) * kBlurTW + lx] + k0 * rows[(ly + 4) * kBlurTW + lx];
            int v = (acc + (1 << 15)) >> 16;
            v = min(max(v, 0), 255);
            blur[(size_t)slice * G->roi_bytes + L.roi_off + (size_t)y * L.w + x] = (uint8_t)v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// rBRIEF: 32 lanes per keypoint, lane b computes descriptor byte b (16 taps)
// computeOrbDescriptor (:108-157) for one descriptor byte; img = blurred level (w x h, step = w, no border)
__device__ __forceinline__ int brief_byte(const uint8_t* __restrict__ img, int w, int h, int cx, int cy, float angle_deg, int b, int& oob)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = angle_deg * factorPI;
    float sb, ca;
    dev_sincosf(angle, &sb, &ca);
    const float a = ca, bb = sb;
    const int step = w;
    const int base = cy * step + cx;
    const int total = w * h;
    const signed char* pat = c_pattern + b * 32;
    int val = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int t[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const float px = (float)pat[(2 * k + s) * 2], py = (float)pat[(2 * k + s) * 2 + 1];
            const float r0 = px * bb, r1 = py * a;
            const float c0 = px * a, c1 = py * bb;
            const int dy = dev_cvround(r0 + r1);
            const int dx = dev_cvround(c0 - c1);
            const int idx = base + dy * step + dx;
            if (idx < 0 || idx >= total) { oob = 1; t[s] = 0; }
            else t[s] = img[idx];
        }
        val |= (t[0] < t[1]) << k;
    }
    return val;
}

__global__ __launch_bounds__(256) void brief_kernel(const DevGeom* __restrict__ G, const uint8_t* __restrict__ blur,
                                                    const uint32_t* __restrict__ lvl_kp, const int32_t* __restrict__ lvl_cnt,
                                                    const float* __restrict__ kp_angle, uint8_t* __restrict__ lvl_desc,
                                                    uint8_t* __restrict__ lvl_oob)
{
    const int slice = blockIdx.y;
    const int gid = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int b = threadIdx.x & 31;
    if (gid >= G->kp_total) return;
    int level = 0;
    while (level + 1 < G->nlevels && gid >= G->lv[level + 1].kp_off) level++;
    const LevelGeom& L = G->lv[level];
    const int i = gid - L.kp_off;
    if (i >= lvl_cnt[slice * G->nlevels + level]) return;
    const uint32_t p = lvl_kp[(size_t)slice * G->kp_total + gid];
    const int cx = (int)(p & 0xfff) + L.minBX, cy = (int)((p >> 12) & 0xfff) + L.minBY;
    const uint8_t* img = blur + (size_t)slice * G->roi_bytes + L.roi_off;
    int oob = 0;
    const int val = brief_byte(img, L.w, L.h, cx, cy, kp_angle[(size_t)slice * G->kp_total + gid], b, oob);
    lvl_desc[((size_t)slice * G->kp_total + gid) * 32 + b] = (uint8_t)val;
    const uint64_t anyoob = __ballot(oob != 0);
    if (b == 0) {
        const int half = (threadIdx.x >> 5) & 1;
        const uint32_t m = half ? (uint32_t)(anyoob >> 32) : (uint32_t)anyoob;
        lvl_oob[(size_t)slice * G->kp_total + gid] = m ? 1 : 0;
    }
}

// ORBextractor::ComputeTrackedKPtsDesc (:1316-1363) [mode 0] and AssignKPtLevelByBestDesc (:1267-1314) [mode 1]:
// 32 lanes per tracked keypoint.  mode 0: descriptor at the keypoint's octave; mode 1: descriptor at every level, the
// level with the smallest Hamming distance to ref_desc (first minimum, strict <) becomes the octave.
__global__ __launch_bounds__(256) void tracked_desc_kernel(const DevGeom* __restrict__ G, const uint8_t* __restrict__ blur,
                                                           eorb_keypoint* __restrict__ kps, int n, int mode,
                                                           const uint8_t* __restrict__ ref_desc, uint8_t* __restrict__ desc,
                                                           uint8_t* __restrict__ oobf)
{
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int b = threadIdx.x & 31;
    if (i >= n) return;
    const eorb_keypoint kp = kps[i];
    const int half = (threadIdx.x >> 5) & 1;
    if (mode == 0) {
        const int level = kp.octave;
        int val = 0, oob = 0;
        if (level >= 0 && level < G->nlevels) {
            const LevelGeom& L = G->lv[level];
            const float scale = 1.0f / G->sf[level];                                  // mvInvScaleFactor[level] (:436)
            const int cx = dev_cvround(kp.x * scale), cy = dev_cvround(kp.y * scale);
            val = brief_byte(blur + L.roi_off, L.w, L.h, cx, cy, kp.angle, b, oob);
        }
        desc[(size_t)i * 32 + b] = (uint8_t)val;
        const uint64_t anyoob = __ballot(oob != 0);
        if (b == 0 && oobf) oobf[i] = (half ? (uint32_t)(anyoob >> 32) : (uint32_t)anyoob) ? 1 : 0;
    } else {
        const int refb = ref_desc[(size_t)i * 32 + b];
        int minDist = 0x7fffffff, best = kp.octave;
        for (int level = 0; level < G->nlevels; level++) {
            const LevelGeom& L = G->lv[level];
            const float scale = 1.0f / G->sf[level];
            const int cx = dev_cvround(kp.x * scale), cy = dev_cvround(kp.y * scale);
            int oob = 0;
            const int val = brief_byte(blur + L.roi_off, L.w, L.h, cx, cy, kp.angle, b, oob);
            int d = __popc((unsigned)(val ^ refb));
#pragma unroll
            for (int sft = 16; sft >= 1; sft >>= 1) d += __shfl_xor(d, sft, 64);   // sum over the 32 lanes of this keypoint
            if (d < minDist) { minDist = d; best = level; }
        }
        if (b == 0) kps[i].octave = best;
    }
}

// ---------------------------------------------------------------------------------------------------
// output ordering of operator() :1150-1173 / :1207-1236
__global__ __launch_bounds__(256) void assemble_kernel(const DevGeom* __restrict__ G, const uint32_t* __restrict__ lvl_kp,
                                                       const int32_t* __restrict__ lvl_cnt, const float* __restrict__ kp_angle,
                                                       const uint8_t* __restrict__ lvl_desc, const uint8_t* __restrict__ lvl_oob,
                                                       int lap0, int lap1, int want_desc, int out_cap,
                                                       eorb_keypoint* __restrict__ out_kp, uint8_t* __restrict__ out_desc,
                                                       uint8_t* __restrict__ out_oob, int32_t* __restrict__ out_n,
                                                       int32_t* __restrict__ out_mono, const int32_t* __restrict__ err_flag,
                                                       int32_t* __restrict__ out_flag)
{
    __shared__ int s_w[4];
    const int slice = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (out_flag && slice == 0 && tid == 0) *out_flag = *err_flag;         // (host entry point: the flag travels with the counters)
    int nk = 0;
    for (int l = 0; l < G->nlevels; l++) nk += lvl_cnt[slice * G->nlevels + l];
    if (tid == 0) out_n[slice] = nk;
    int mono_run = 0, stereo_run = 0;      // counts of each class before the current pass
    int e0 = 0;                            // emission index of the first keypoint of the current level
    for (int l = 0; l < G->nlevels; l++) {
        const LevelGeom& L = G->lv[l];
        const int nl = lvl_cnt[slice * G->nlevels + l];
        const float scale = G->sf[l];
        for (int i0 = 0; i0 < nl; i0 += 256) {
            const int i = i0 + tid;
            bool valid = i < nl, st = false;
            eorb_keypoint kp{};
            size_t gi = 0;
            if (valid) {
                gi = (size_t)slice * G->kp_total + L.kp_off + i;
                const uint32_t p = lvl_kp[gi];
                kp.x = (float)(int)(p & 0xfff) + (float)L.minBX;
                kp.y = (float)(int)((p >> 12) & 0xfff) + (float)L.minBY;
                if (l != 0) { kp.x = kp.x * scale; kp.y = kp.y * scale; }       // keypoint.pt *= scale
                kp.size = (float)L.patch_size;
                kp.angle = kp_angle[gi];
                kp.response = (float)(p >> 24);
                kp.octave = l; kp.class_id = -1;
                st = kp.x >= (float)lap0 && kp.x <= (float)lap1;
            }
            const uint64_t bs = __ballot(valid && st), bm = __ballot(valid && !st);
            if (lane == 0) s_w[w] = (int)__popcll(bs) | ((int)__popcll(bm) << 16);
            __syncthreads();
            int sb = stereo_run, mb = mono_run, ts = 0, tm = 0;
            for (int k = 0; k < 4; k++) {
                const int cs = s_w[k] & 0xffff, cm = s_w[k] >> 16;
                if (k < w) { sb += cs; mb += cm; }
                ts += cs; tm += cm;
            }
            if (valid) {
                const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                int dst;
                if (st) dst = nk - 1 - (sb + (int)__popcll(bs & lt));
                else dst = mb + (int)__popcll(bm & lt);
                if (dst >= 0 && dst < out_cap) {
                    out_kp[(size_t)slice * out_cap + dst] = kp;
                    if (want_desc && out_desc) {
                        const uint4* s4 = (const uint4*)(lvl_desc + gi * 32);
                        uint4* d4 = (uint4*)(out_desc + ((size_t)slice * out_cap + dst) * 32);
                        d4[0] = s4[0]; d4[1] = s4[1];
                    }
                    if (out_oob) out_oob[(size_t)slice * out_cap + dst] = want_desc ? lvl_oob[gi] : 0;
                }
            }
            stereo_run += ts; mono_run += tm;
            __syncthreads();
        }
        e0 += nl;
    }
    (void)e0;
    if (tid == 0 && out_mono) out_mono[slice] = mono_run;
}

// ---------------------------------------------------------------------------------------------------
// orientation, descriptor and output record of a keypoint by its 32 lanes in ONE launch: orient_kernel + brief_kernel + assemble_kernel for
// the calls whose lapping area holds no keypoint (all "mono": they leave level by level in the octree's order, slot i of level l goes to
// e = sum(cnt[l' < l]) + i, :1150-1173) or every keypoint (all_stereo, the monocular pipeline's (0, 1000): filled from the back, n - 1 - e,
// monoIndex 0).  Three dependent launches cost a one-frame call 3 us each and their own ramps; the arithmetic is the
// kernels' own (IC_Angle :77-104 as integer moments, computeOrbDescriptor :108-157 through brief_byte).
template <bool DESC>
__global__ __launch_bounds__(256) void describe_kernel(const DevGeom* __restrict__ G, const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                       const uint32_t* __restrict__ lvl_kp, const int32_t* __restrict__ lvl_cnt, int all_stereo, int out_cap,
                                                       eorb_keypoint* __restrict__ out_kp, uint8_t* __restrict__ out_desc, uint8_t* __restrict__ out_oob,
                                                       int32_t* __restrict__ out_n, int32_t* __restrict__ out_mono, const int32_t* __restrict__ err_flag,
                                                       int32_t* __restrict__ out_flag)
{
    const int slice = blockIdx.y;
    const int gid = blockIdx.x * 8 + (threadIdx.x >> 5);       // keypoint slot in the slice's level arrays
    const int l32 = (int)(threadIdx.x & 31);
    const int32_t* cnt = lvl_cnt + slice * G->nlevels;
    int nk = 0;
    for (int l = 0; l < G->nlevels; l++) nk += cnt[l];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out_n[slice] = nk;
        if (out_mono) out_mono[slice] = all_stereo ? 0 : nk;
        if (out_flag && slice == 0) *out_flag = *err_flag;      // (host entry point: the flag travels with the counters)
    }
    if (gid >= G->kp_total) return;
    int level = 0, before = 0;
    while (level + 1 < G->nlevels && gid >= G->lv[level + 1].kp_off) { before += cnt[level]; level++; }
    const LevelGeom& L = G->lv[level];
    const int i = gid - L.kp_off;
    if (i >= cnt[level]) return;                               // whole 32-lane group exits together
    const uint32_t p = lvl_kp[(size_t)slice * G->kp_total + gid];
    const int cx = (int)(p & 0xfff) + L.minBX, cy = (int)((p >> 12) & 0xfff) + L.minBY;
    // ---- IC_Angle: lane = column u of the disc (-15 .. 15, one lane idle) ----
    const uint8_t* center = pyr + (size_t)slice * G->pyr_bytes + L.buf_off + (size_t)(cy + G->edge) * L.bw + (cx + G->edge);
    const int step = L.bw;
    const int u = l32 - 15;
    const int au = u < 0 ? -u : u;
    int m_01 = 0, m_10 = 0;
    if (au <= 15) m_10 = u * (int)center[u];
#pragma unroll
    for (int v = 1; v <= 15; ++v) {
        if (au <= G->umax[v]) {
            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
            m_01 += v * (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {
        m_01 += __shfl_xor(m_01, d, 64);
        m_10 += __shfl_xor(m_10, d, 64);
    }
    const float angle = dev_fast_atan2((float)m_01, (float)m_10);          // (every lane of the group holds both sums)
    const int dst = all_stereo ? nk - 1 - (before + i) : before + i;
    int val = 0, oob = 0;
    if (DESC) val = brief_byte(blur + (size_t)slice * G->roi_bytes + L.roi_off, L.w, L.h, cx, cy, angle, l32, oob);
    const uint64_t anyoob = __ballot(oob != 0);
    if (dst < 0 || dst >= out_cap) return;
    if (DESC && out_desc) out_desc[((size_t)slice * out_cap + dst) * 32 + l32] = (uint8_t)val;
    if (l32 == 0) {
        eorb_keypoint kp{};
        kp.x = (float)(int)(p & 0xfff) + (float)L.minBX;
        kp.y = (float)(int)((p >> 12) & 0xfff) + (float)L.minBY;
        const float scale = G->sf[level];
        if (level != 0) { kp.x = kp.x * scale; kp.y = kp.y * scale; }       // keypoint.pt *= scale
        kp.size = (float)L.patch_size;
        kp.angle = angle;
        kp.response = (float)(p >> 24);
        kp.octave = level; kp.class_id = -1;
        out_kp[(size_t)slice * out_cap + dst] = kp;
        if (out_oob) {
            const uint32_t m = ((threadIdx.x >> 5) & 1) ? (uint32_t)(anyoob >> 32) : (uint32_t)anyoob;
            out_oob[(size_t)slice * out_cap + dst] = (DESC && m) ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side: ORBextractor::ORBextractor (src/ORBextractor.cc:420-489) + per-size geometry
static int cv_round_d(double v) { return (int)lrint(v); }

int orb_configure(eorb_ctx* c, const eorb_orb_params* p, int W, int H)
{
    OrbState& o = c->orb;
    o.configured = false;
    if (!p || p->nlevels < 1 || p->nlevels > kMaxLevels || W <= 0 || H <= 0 || p->nfeatures < 0)
        return set_err(c, EORB_E_ARG, "orb_configure: bad parameters");
    o.p = *p; o.W = W; o.H = H; o.nlevels = p->nlevels;
    const int nlevels = p->nlevels;
    const double scaleFactor = (double)p->scaleFactor;
    o.sf[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) o.sf[i] = (float)((double)o.sf[i - 1] * scaleFactor);
    for (int i = 0; i < nlevels; i++) o.inv_sf[i] = 1.0f / o.sf[i];
    {
        const float factor = (float)(1.0 / scaleFactor);
        float nDesired = (float)p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; l++) {
            o.nfeat[l] = cv_round_d((double)nDesired);
            sum += o.nfeat[l];
            nDesired *= factor;
        }
        o.nfeat[nlevels - 1] = std::max(p->nfeatures - sum, 0);
    }
    {
        int v, v0, vmax = (int)floor(15 * sqrt(2.f) / 2 + 1);
        int vmin = (int)ceil(15 * sqrt(2.f) / 2);
        const double hp2 = 15 * 15;
        for (v = 0; v <= vmax; ++v) o.umax[v] = cv_round_d(sqrt(hp2 - v * v));
        for (v = 15, v0 = 0; v >= vmin; --v) {
            while (o.umax[v0] == o.umax[v0 + 1]) ++v0;
            o.umax[v] = v0;
            ++v0;
        }
    }
    if (p->edgeTh < 0) {
        float ne = 19 * ((float)p->imWidth / (float)752);
        o.edge = (int)ne;
        o.edge += (o.edge % 2 - 1);
    } else o.edge = p->edgeTh;
    const int E = o.edge;
    if (E < 8) return set_err(c, EORB_E_CONFIG, "edge threshold %d < 8: the orientation patch would leave the bordered level buffer", E);

    std::vector<short> tabs;     // short4 entries
    int pyr = 0, roi = 0, cells = 0, cell_cap = 0, cand_total = 0, kp_total = 0;
    int ncap_max = 0, node_cap_max = 0, vsp_cap_max = 0;
    for (int l = 0; l < nlevels; l++) {
        LevelGeom& L = o.lv[l];
        const float scale = o.inv_sf[l];
        L.w = cv_round_d((double)((float)W * scale)); L.h = cv_round_d((double)((float)H * scale));
        if (L.w < 1 || L.h < 1) return set_err(c, EORB_E_CONFIG, "level %d is empty", l);
        L.bw = L.w + 2 * E; L.bh = L.h + 2 * E;
        L.buf_off = pyr; pyr += (L.bw * L.bh + 63) & ~63;
        L.roi_off = roi; roi += (L.w * L.h + 63) & ~63;
        L.minBX = E - 3; L.minBY = E - 3; L.maxBX = L.w - E + 3; L.maxBY = L.h - E + 3;
        const float width = (float)(L.maxBX - L.minBX), height = (float)(L.maxBY - L.minBY);
        L.nCols = (int)(width / 30.0f); L.nRows = (int)(height / 30.0f);
        if (L.nCols < 1 || L.nRows < 1)          // the reference divides by zero here (SURVEY App.B H14)
            return set_err(c, EORB_E_CONFIG, "level %d (%dx%d) is smaller than one 30-px cell with edge %d", l, L.w, L.h, E);
        L.wCell = (int)ceilf(width / (float)L.nCols); L.hCell = (int)ceilf(height / (float)L.nRows);
        if (L.wCell > kCellMax || L.hCell > kCellMax) return set_err(c, EORB_E_CONFIG, "cell larger than %d px", kCellMax);
        const int nIni = (int)roundf(width / height);
        if (nIni < 1) return set_err(c, EORB_E_CONFIG, "level %d: portrait aspect gives zero octree roots", l);
        if (L.maxBX - L.minBX >= 4096 || L.maxBY - L.minBY >= 4096) return set_err(c, EORB_E_CONFIG, "level too large");
        L.cell_off = cells; cells += L.nCols * L.nRows;
        cell_cap = std::max(cell_cap, ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2));
        L.nfeat = o.nfeat[l];
        L.kp_cap = L.nfeat + 8;
        L.kp_off = kp_total; kp_total += L.kp_cap;
        L.scale = o.sf[l];
        L.patch_size = (int)(31.0f * o.sf[l]);
        L.node_cap = L.nfeat + 4 * nIni + 24;
        node_cap_max = std::max(node_cap_max, L.node_cap);
        vsp_cap_max = std::max(vsp_cap_max, 2 * (L.nfeat + 16));
        // resize tables (cv::resize INTER_LINEAR 8u, SURVEY App.B H5)
        L.xtab_off = (int)tabs.size() / 4;
        L.xmax = L.w;
        if (l > 0) {
            const LevelGeom& P = o.lv[l - 1];
            const double inv_scale_x = (double)L.w / P.w, inv_scale_y = (double)L.h / P.h;
            const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
            if (P.w == 2 * L.w && P.h == 2 * L.h)
                return set_err(c, EORB_E_CONFIG, "exact 2x decimation switches cv::resize to INTER_AREA: unsupported");
            for (int dx = 0; dx < L.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = (int)floor(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx + 1 >= P.w) { L.xmax = std::min(L.xmax, dx); if (sx >= P.w - 1) { fx = 0; sx = P.w - 1; } }
                const int a0 = cv_round_d((double)((1.f - fx) * 2048)), a1 = cv_round_d((double)(fx * 2048));
                tabs.push_back((short)sx); tabs.push_back((short)a0); tabs.push_back((short)a1); tabs.push_back(0);
            }
            L.ytab_off = (int)tabs.size() / 4;
            for (int dy = 0; dy < L.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = (int)floor(fy);
                fy -= sy;
                const int b0 = cv_round_d((double)((1.f - fy) * 2048)), b1 = cv_round_d((double)(fy * 2048));
                const int sy0 = sy < 0 ? 0 : (sy < P.h ? sy : P.h - 1);
                const int sy1 = sy + 1 < 0 ? 0 : (sy + 1 < P.h ? sy + 1 : P.h - 1);
                tabs.push_back((short)sy0); tabs.push_back((short)sy1); tabs.push_back((short)b0); tabs.push_back((short)b1);
            }
        } else {
            L.ytab_off = L.xtab_off;
        }
    }
    for (int l = 0; l < nlevels; l++) {
        LevelGeom& L = o.lv[l];
        // after the 3x3 non-maximum suppression at most every other pixel of a cell (in x and in y) survives: the level's own
        // cell size bounds its candidates (cell_cap, the largest cell of ANY level, is only the stride of the per-cell arrays)
        L.cand_cap = L.nCols * L.nRows * (((L.wCell + 1) / 2) * ((L.hCell + 1) / 2));
        L.cand_off = cand_total; cand_total += L.cand_cap;
        ncap_max = std::max(ncap_max, L.cand_cap);
    }
    // The octree addresses a level's candidates with 16-bit keys.  cand_cap is the worst case (every other pixel in x and in y a
    // corner: 90 000 on level 0 of the reference's default 752x480 frames, include/ORBextractor.h:31, Examples/Event/EuRoC.yaml:86);
    // real frames stay far below it, so the working arrays hold at most 65 534 candidates per level and a level that does exceed
    // them is truncated AND reported (status bit 1 of the call / eorb_sync), not refused up front.
    ncap_max = std::min(ncap_max, 65534);
    if (node_cap_max > 16000) return set_err(c, EORB_E_CAPACITY, "more than 16 000 features on one pyramid level");   // 16-bit packed child counters
    o.pyr_bytes = pyr; o.roi_bytes = roi; o.ncells = cells; o.cell_cap = cell_cap;
    o.cand_total = cand_total; o.kp_total = kp_total; o.max_out = kp_total;
    // octree working set: greedy placement into the 160 KB of LDS (most latency-critical first), the rest into a per-(slice,
    // level) block of global scratch (346x260 with 3 000 features on one level, VGA-class frames)
    if (c->dbg_pool_shrink > 0)                    // test hook: force node-pool overflows (sticky status of the *_dev paths)
        for (int l = 0; l < nlevels; l++) o.lv[l].node_cap = std::max(8, o.lv[l].node_cap - c->dbg_pool_shrink);
    // vsp lists are sorted with a bitonic network: capacity rounded up to a power of two
    int vsp_pow2 = 1; while (vsp_pow2 < vsp_cap_max) vsp_pow2 <<= 1;
    const size_t item_bytes[7] = {(size_t)2 * node_cap_max * 16,
                                  sizeof(uint64_t) * (size_t)vsp_pow2, sizeof(uint64_t) * (size_t)vsp_pow2,
                                  (sizeof(uint16_t) * (size_t)ncap_max + 15) & ~(size_t)15, (sizeof(uint16_t) * (size_t)ncap_max + 15) & ~(size_t)15,
                                  (sizeof(uint32_t) * (size_t)ncap_max + 15) & ~(size_t)15,
                                  ((size_t)node_cap_max * (4 + 2 + 2 + 2 + 1 + 8) + 8 + 15) & ~(size_t)15};
    int oct_in_lds[2][7], oct_off[2][7];
    // placement order: the per-round arrays and the node arrays first (touched by every step), then the size lists, keys, points
    const int order[7] = {6, 0, 1, 2, 3, 4, 5};
    static const int budget_kb = [] { const char* e = getenv("EORB_OCT_BUDGET_KB"); return e ? atoi(e) : 156; }();      // (A/B runs)
    for (int v = 0; v < 2; v++) {
        // measured on 1 024 frames of 240x180 / 400 features: 344 us with 150 KB per workgroup, 211 us with 76 KB, 253 us with the
        // whole working set in global memory; a single frame per call is 5 % faster with everything in LDS
        const size_t lds_budget = c->dbg_force_global ? 0 : (v == 0 ? (size_t)budget_kb * 1024 : std::min<size_t>((size_t)budget_kb, 76) * 1024);
        size_t lds = 0, gblock = 0;
        for (int oi = 0; oi < 7; oi++) {
            const int k = order[oi];
            if (lds + item_bytes[k] <= lds_budget) { oct_in_lds[v][k] = 1; oct_off[v][k] = (int)lds; lds += item_bytes[k]; }
            else { oct_in_lds[v][k] = 0; oct_off[v][k] = (int)gblock; gblock += item_bytes[k]; }
        }
        o.oct_lds[v] = (int)lds;
        o.oct_scratch[v] = (int)gblock;
    }

    DevGeom g{};
    memcpy(g.lv, o.lv, sizeof(o.lv));
    g.nlevels = nlevels; g.edge = E; g.W = W; g.H = H;
    g.pyr_bytes = pyr; g.roi_bytes = roi; g.ncells = cells; g.cell_cap = cell_cap; g.cand_total = cand_total;
    g.kp_total = kp_total; g.max_out = kp_total; g.iniTh = std::min(std::max(p->iniThFAST, 0), 255);
    g.minTh = std::min(std::max(p->minThFAST, 0), 255);
    memcpy(g.umax, o.umax, sizeof(o.umax)); memcpy(g.sf, o.sf, sizeof(o.sf));
    memcpy(g.oct_in_lds, oct_in_lds, sizeof(oct_in_lds)); memcpy(g.oct_off, oct_off, sizeof(oct_off));
    for (int v = 0; v < 2; v++) {
        g.oct_lds_bytes[v] = o.oct_lds[v]; g.oct_gblock_bytes[v] = o.oct_scratch[v];
        bool all = true; for (int k = 0; k < 7; k++) all = all && oct_in_lds[v][k];
        static const int direct_on = [] { const char* e = getenv("EORB_OCT_DIRECT"); return e ? atoi(e) : 1; }();       // (A/B runs, parity tests of the list algorithm)
        // the dynamic placement: the items sized by the features (6, 0, 1, 2) at fixed LDS offsets, the rest behind them
        {
            const size_t lds_budget = c->dbg_force_global ? 0 : (v == 0 ? (size_t)budget_kb * 1024 : std::min<size_t>((size_t)budget_kb, 76) * 1024);
            size_t lds = 0;
            const int fixed[4] = {6, 0, 1, 2};
            for (int k = 0; k < 7; k++) g.oct_off_dyn[v][k] = 0;
            for (int fi = 0; fi < 4; fi++) { g.oct_off_dyn[v][fixed[fi]] = (int)lds; lds += item_bytes[fixed[fi]]; }
            g.oct_dyn_base[v] = (int)lds; g.oct_dyn_lds[v] = (int)lds_budget;
            static const int dyn_on = [] { const char* e = getenv("EORB_OCT_DYNAMIC"); return e ? atoi(e) : 1; }();
            // (in use when the static placement leaves something in global memory and the fixed items leave room for 1 024 candidates)
            g.oct_dyn[v] = (dyn_on && !all && lds + 8 * 1024 <= lds_budget) ? ((direct_on && !c->dbg_oct_list) ? std::min(4096, 2 * vsp_pow2) : 1) : 0;
            if (g.oct_dyn[v] && !(direct_on && !c->dbg_oct_list)) g.oct_dyn[v] = -1;     // (dynamic, no direct passes: the cap test `n <= -1` fails)
            o.oct_dyn[v] = g.oct_dyn[v]; o.oct_dyn_lds[v] = g.oct_dyn_lds[v];
        }
        o.oct_direct_cap[v] = g.oct_direct_cap[v] = (direct_on && !c->dbg_oct_list && all && oct_off[v][2] == oct_off[v][1] + (int)(sizeof(uint64_t) * vsp_pow2)) ? std::min(4096, 2 * vsp_pow2) : 0;
    }
    g.node_cap_max = node_cap_max; g.vsp_cap_max = vsp_cap_max; g.ncap_max = ncap_max;
    int rc;
    if ((rc = ensure(c, o.geom, sizeof(DevGeom)))) return rc;
    if ((rc = ensure(c, o.tabs, std::max<size_t>(tabs.size() * sizeof(short), 8)))) return rc;
    EORB_HIP(c, hipMemcpy(o.geom.p, &g, sizeof(g), hipMemcpyHostToDevice));
    if (!tabs.empty()) EORB_HIP(c, hipMemcpy(o.tabs.p, tabs.data(), tabs.size() * sizeof(short), hipMemcpyHostToDevice));
    EORB_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), k_orb_pattern_31, 1024));
    EORB_HIP(c, hipFuncSetAttribute((const void*)octree_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    EORB_HIP(c, hipFuncSetAttribute((const void*)octree_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    for (int v = 0; v < 2; v++) { o.oct_all_lds[v] = 1; for (int k = 0; k < 7; k++) o.oct_all_lds[v] &= oct_in_lds[v][k]; }
    o.configured = true;
    return EORB_OK;
}

int orb_extract_dev(eorb_ctx* c, const uint8_t* d_img, int img_stride, size_t img_slice_bytes, int B, int lap0, int lap1,
                    int want_desc, eorb_keypoint* d_kps, uint8_t* d_desc, uint8_t* d_oob, int32_t* d_n, int32_t* d_mono,
                    int32_t* d_flag_out)
{
    OrbState& o = c->orb;
    if (!o.configured) return set_err(c, EORB_E_NOTCONF, "orb_extract: eorb_orb_configure not called");
    if (B <= 0) return EORB_OK;
    int rc;
    const size_t nb = (size_t)B;
    if ((rc = ensure(c, c->pyr, nb * o.pyr_bytes))) return rc;
    if ((rc = ensure(c, c->blur, nb * o.roi_bytes))) return rc;
    if ((rc = ensure(c, c->cell_cnt, nb * o.ncells * sizeof(int32_t)))) return rc;
    if ((rc = ensure(c, c->cell_cand, nb * o.ncells * (size_t)o.cell_cap * sizeof(uint32_t)))) return rc;
    const int placement = B * o.nlevels > 256 ? 1 : 0;                                         // more workgroups than CUs: two per CU
    if ((rc = ensure(c, c->oct_scratch, nb * o.nlevels * (size_t)o.oct_scratch[placement]))) return rc;   // octree items that do not fit the LDS
    if ((rc = ensure(c, c->lvl_kp, nb * o.kp_total * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->lvl_cnt, 2 * nb * o.nlevels * sizeof(int32_t) + 64))) return rc;     // counts | error flag (16 ints) | the octree's redo flags
    if ((rc = ensure(c, c->kp_angle, nb * o.kp_total * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->out_desc, nb * o.kp_total * 32))) return rc;                        // per-level descriptors
    if ((rc = ensure(c, c->out_oob, nb * o.kp_total))) return rc;
    const DevGeom* G = (const DevGeom*)o.geom.p;
    uint8_t* pyr = (uint8_t*)c->pyr.p;
    int32_t* err_flag = (int32_t*)((char*)c->lvl_cnt.p + nb * o.nlevels * sizeof(int32_t));
    {
        ProfScope ps(c, "orb_pyr");
        const int n0 = o.lv[0].bw * o.lv[0].bh;
        if (c->pyr0_f32) {
            // (the caller left the normalisation of its float images to this kernel; d_img is where the u8 images go)
            pyr_level0_f32_kernel<<<dim3((n0 / 4 + 256) / 256, B), 256, 0, c->stream>>>(c->pyr0_f32, c->pyr0_mm, const_cast<uint8_t*>(d_img), img_stride, img_slice_bytes, G, pyr, err_flag);
            c->pyr0_f32 = nullptr; c->pyr0_mm = nullptr;
        } else
        pyr_level0_kernel<<<dim3((n0 / 4 + 256) / 256, B), 256, 0, c->stream>>>(d_img, img_stride, img_slice_bytes, G, pyr, err_flag);
        for (int l = 1; l < o.nlevels; l++) {
            const int n = o.lv[l].bw * o.lv[l].bh;
            pyr_resize_kernel<<<dim3(B >= 16 ? (n + 1023) / 1024 : (n + 255) / 256, B), 256, 0, c->stream>>>(l, G, (const short4*)o.tabs.p, pyr);
        }
        EORB_LAUNCH_CHECK(c, "pyramid kernels");
    }
    {
        ProfScope ps(c, "orb_fast_cells");
        fast_cells_kernel<<<dim3(o.ncells, B), 256, 0, c->stream>>>(G, pyr, (int32_t*)c->cell_cnt.p, (uint32_t*)c->cell_cand.p);
        EORB_LAUNCH_CHECK(c, "fast_cells_kernel");
    }
    {
        ProfScope ps(c, "orb_octree");
        // a call that hands its own flag back (the host entry point) stays out of the sticky word of the *_dev calls: eorb_sync still
        // reports an earlier batch's overflow
        int32_t* sticky = d_flag_out ? nullptr : (int32_t*)c->status.p;
        int32_t* redo = err_flag + 16;                               // one flag per (slice, level), behind the error flag
        if (o.oct_all_lds[placement])
            octree_kernel<true><<<B * o.nlevels, kOctThreads, o.oct_lds[placement], c->stream>>>(G, (const int32_t*)c->cell_cnt.p, (const uint32_t*)c->cell_cand.p,
                                                                    (unsigned char*)c->oct_scratch.p, (uint32_t*)c->lvl_kp.p, (int32_t*)c->lvl_cnt.p, err_flag, sticky, placement, nullptr);
        else if (o.oct_dyn[placement]) {
            // levels whose candidates fit the LDS this time run there; the others are left to the mixed placement
            octree_kernel<true><<<B * o.nlevels, kOctThreads, o.oct_dyn_lds[placement], c->stream>>>(G, (const int32_t*)c->cell_cnt.p, (const uint32_t*)c->cell_cand.p,
                                                                    (unsigned char*)c->oct_scratch.p, (uint32_t*)c->lvl_kp.p, (int32_t*)c->lvl_cnt.p, err_flag, sticky, placement, redo);
            octree_kernel<false><<<B * o.nlevels, kOctThreads, o.oct_lds[placement], c->stream>>>(G, (const int32_t*)c->cell_cnt.p, (const uint32_t*)c->cell_cand.p,
                                                                    (unsigned char*)c->oct_scratch.p, (uint32_t*)c->lvl_kp.p, (int32_t*)c->lvl_cnt.p, err_flag, sticky, placement, redo);
        } else
            octree_kernel<false><<<B * o.nlevels, kOctThreads, o.oct_lds[placement], c->stream>>>(G, (const int32_t*)c->cell_cnt.p, (const uint32_t*)c->cell_cand.p,
                                                                    (unsigned char*)c->oct_scratch.p, (uint32_t*)c->lvl_kp.p, (int32_t*)c->lvl_cnt.p, err_flag, sticky, placement, nullptr);
        EORB_LAUNCH_CHECK(c, "octree_kernel");
    }
    // a lapping area [lap0, lap1] that ends left of the first column a keypoint can have holds none of them, one that spans the image
    // (the monocular pipeline's (0, 1000)) all of them: one launch then does orientation, descriptor and output order (describe_kernel)
    static const int fuse_env = [] { const char* e = getenv("EORB_ORB_DESCRIBE"); return e ? atoi(e) : 1; }();      // (A/B runs)
    const bool none_in = (float)lap1 < (float)o.lv[0].minBX, all_in = lap0 <= 0 && lap1 >= o.W;
    const bool one_launch = fuse_env != 0 && !c->dbg_orb_three_launches && (none_in || all_in);
    const int all_stereo = !none_in && all_in;
    if (want_desc) {
        ProfScope ps(c, "orb_blur");
        int tyt = 0, wmax = 0;
        for (int l = 0; l < o.nlevels; l++) { tyt += (o.lv[l].h + kBlurTH - 1) / kBlurTH; wmax = std::max(wmax, o.lv[l].w); }
        blur_kernel<<<dim3((wmax + kBlurTW - 1) / kBlurTW, tyt, B), 256, 0, c->stream>>>(G, pyr, (uint8_t*)c->blur.p);
        EORB_LAUNCH_CHECK(c, "blur_kernel");
    }
    if (one_launch) {
        ProfScope ps(c, "orb_describe");
        const dim3 grid((o.kp_total + 7) / 8, B);
        if (want_desc)
            describe_kernel<true><<<grid, 256, 0, c->stream>>>(G, pyr, (const uint8_t*)c->blur.p, (const uint32_t*)c->lvl_kp.p, (const int32_t*)c->lvl_cnt.p,
                                                              all_stereo, o.max_out, d_kps, d_desc, d_oob, d_n, d_mono, err_flag, d_flag_out);
        else
            describe_kernel<false><<<grid, 256, 0, c->stream>>>(G, pyr, nullptr, (const uint32_t*)c->lvl_kp.p, (const int32_t*)c->lvl_cnt.p,
                                                               all_stereo, o.max_out, d_kps, d_desc, d_oob, d_n, d_mono, err_flag, d_flag_out);
        EORB_LAUNCH_CHECK(c, "describe_kernel");
        return EORB_OK;
    }
    {
        ProfScope ps(c, "orb_orient");
        orient_kernel<<<dim3((o.kp_total + 7) / 8, B), 256, 0, c->stream>>>(G, pyr, (const uint32_t*)c->lvl_kp.p,
                                                                              (const int32_t*)c->lvl_cnt.p, (float*)c->kp_angle.p);
        EORB_LAUNCH_CHECK(c, "orient_kernel");
    }
    if (want_desc) {
        ProfScope ps(c, "orb_brief");
        brief_kernel<<<dim3((o.kp_total + 7) / 8, B), 256, 0, c->stream>>>(G, (const uint8_t*)c->blur.p, (const uint32_t*)c->lvl_kp.p,
                                                                           (const int32_t*)c->lvl_cnt.p, (const float*)c->kp_angle.p,
                                                                           (uint8_t*)c->out_desc.p, (uint8_t*)c->out_oob.p);
        EORB_LAUNCH_CHECK(c, "brief_kernel");
    }
    {
        ProfScope ps(c, "orb_assemble");
        assemble_kernel<<<B, 256, 0, c->stream>>>(G, (const uint32_t*)c->lvl_kp.p, (const int32_t*)c->lvl_cnt.p,
                                                  (const float*)c->kp_angle.p, (const uint8_t*)c->out_desc.p,
                                                  (const uint8_t*)c->out_oob.p, lap0, lap1, want_desc, o.max_out, d_kps, d_desc,
                                                  d_oob, d_n, d_mono, err_flag, d_flag_out);
        EORB_LAUNCH_CHECK(c, "assemble_kernel");
    }
    return EORB_OK;
}

// pyramid + blurred planes of ONE image (the front half of operator(), shared by the tracked-keypoint helpers)
int orb_pyramid_blur_dev(eorb_ctx* c, const uint8_t* d_img, int img_stride)
{
    OrbState& o = c->orb;
    if (!o.configured) return set_err(c, EORB_E_NOTCONF, "orb: eorb_orb_configure not called");
    int rc;
    if ((rc = ensure(c, c->pyr, (size_t)o.pyr_bytes))) return rc;
    if ((rc = ensure(c, c->blur, (size_t)o.roi_bytes))) return rc;
    const DevGeom* G = (const DevGeom*)o.geom.p;
    uint8_t* pyr = (uint8_t*)c->pyr.p;
    ProfScope ps(c, "orb_pyr_blur");
    const int n0 = o.lv[0].bw * o.lv[0].bh;
    pyr_level0_kernel<<<dim3((n0 + 255) / 256, 1), 256, 0, c->stream>>>(d_img, img_stride, 0, G, pyr, nullptr);
    for (int l = 1; l < o.nlevels; l++) {
        const int n = o.lv[l].bw * o.lv[l].bh;
        pyr_resize_kernel<<<dim3((n + 255) / 256, 1), 256, 0, c->stream>>>(l, G, (const short4*)o.tabs.p, pyr);
    }
    int tyt = 0, wmax = 0;
    for (int l = 0; l < o.nlevels; l++) { tyt += (o.lv[l].h + kBlurTH - 1) / kBlurTH; wmax = std::max(wmax, o.lv[l].w); }
    blur_kernel<<<dim3((wmax + kBlurTW - 1) / kBlurTW, tyt, 1), 256, 0, c->stream>>>(G, pyr, (uint8_t*)c->blur.p);
    EORB_LAUNCH_CHECK(c, "pyramid/blur kernels");
    return EORB_OK;
}

int orb_tracked_dev(eorb_ctx* c, eorb_keypoint* d_kps, int n, int mode, const uint8_t* d_ref, uint8_t* d_desc, uint8_t* d_oob)
{
    if (n <= 0) return EORB_OK;
    OrbState& o = c->orb;
    ProfScope ps(c, "orb_tracked_desc");
    tracked_desc_kernel<<<(n + 7) / 8, 256, 0, c->stream>>>((const DevGeom*)o.geom.p, (const uint8_t*)c->blur.p, d_kps, n, mode,
                                                           d_ref, d_desc, d_oob);
    EORB_LAUNCH_CHECK(c, "tracked_desc_kernel");
    return EORB_OK;
}

// ---------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (src/Frame.cc:869-1048) on the last extraction of a (left, right) pair = slices 0 and 1 of one batch
// (the reference runs two extractors of equal parameters, :122-125).  One wavefront per left keypoint:
//   candidates (:909-959): every right keypoint whose row band [floor(y - r), ceil(y + r)], r = 2 * scale(octave), holds the row
//   (int)vL, octave within +-1, uR in [uL - mbf / mb, uL]; the smallest descriptor distance, the first in iR order among equals =
//   the wave minimum of (distance << 16 | iR); kept below (TH_HIGH + TH_LOW) / 2 (a best distance in [75, 100) is dropped by :962);
//   correlation (:962-1033): L1 norm of the 11 x 11 patches of the keypoint's level image (mvImagePyramid: the un-blurred level)
//   over the shifts -5..5, lanes = pixels, integer sums; first smallest shift, parabola, re-scaling and disparity in the
//   reference's float operations.  sad[iL] = the norm of an accepted match, -1 otherwise.
// stereo_median_kernel (:1036-1047): the matches ordered by (norm, iL); everything at or above 1.5 * 1.4 * the norm of rank size / 2
// is taken back -- that norm by a two-level counting select, one workgroup.
__global__ __launch_bounds__(256) void stereo_match_kernel(const DevGeom* __restrict__ G, const uint8_t* __restrict__ pyr, const eorb_keypoint* __restrict__ kps,
                                                           const uint8_t* __restrict__ desc, const int32_t* __restrict__ n, int cap, float mb, float mbf,
                                                           float* __restrict__ uRight, float* __restrict__ depth, int32_t* __restrict__ sad)
{
    const int lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int N = min(n[0], cap), Nr = min(n[1], cap);
    if (iL >= N) return;
    const eorb_keypoint kL = kps[iL];
    const eorb_keypoint* kR = kps + cap;
    const uint64_t* dR = (const uint64_t*)(desc + (size_t)cap * 32);
    const uint64_t* dl = (const uint64_t*)(desc + (size_t)iL * 32);
    const uint64_t d0 = dl[0], d1 = dl[1], d2 = dl[2], d3 = dl[3];
    if (lane == 0) { uRight[iL] = -1.0f; depth[iL] = -1.0f; sad[iL] = -1; }
    const int levelL = kL.octave;
    const float vL = kL.y, uL = kL.x;
    const int row = (int)vL;
    const float minD = 0.f, maxD = mbf / mb;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) return;
    uint32_t best = 0xffffffffu;
    for (int i0 = 0; i0 < Nr; i0 += 64) {
        const int iR = i0 + lane;
        uint32_t key = 0xffffffffu;
        if (iR < Nr) {
            const eorb_keypoint k = kR[iR];
            const float r = 2.0f * G->sf[k.octave];
            const int maxr = (int)ceilf(k.y + r), minr = (int)floorf(k.y - r);
            if (row >= minr && row <= maxr && k.octave >= levelL - 1 && k.octave <= levelL + 1 && k.x >= minU && k.x <= maxU) {
                const uint64_t* q = dR + (size_t)iR * 4;
                const int dist = __popcll(d0 ^ q[0]) + __popcll(d1 ^ q[1]) + __popcll(d2 ^ q[2]) + __popcll(d3 ^ q[3]);
                key = ((uint32_t)dist << 16) | (uint32_t)iR;
            }
        }
        best = min(best, key);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, d, 64));
    if (best == 0xffffffffu || (int)(best >> 16) >= (100 + 50) / 2) return;              // TH_HIGH, TH_LOW: src/ORBmatcher.cc:36-37
    const int bestIdxR = (int)(best & 0xffffu);
    const float uR0 = kR[bestIdxR].x;
    const float scaleFactor = 1.0f / G->sf[levelL];                                     // mvInvScaleFactor (src/ORBextractor.cc:436)
    const float scaleduL = roundf(uL * scaleFactor), scaledvL = roundf(vL * scaleFactor), scaleduR0 = roundf(uR0 * scaleFactor);
    const int w = 5, Lw = 5;
    const LevelGeom& LG = G->lv[levelL];
    const float iniu = scaleduR0 + Lw - w, endu = scaleduR0 + Lw + w + 1;
    if (iniu < 0 || endu >= (float)LG.w) return;
    const int r0 = (int)(scaledvL - w), cL0 = (int)(scaleduL - w);
    // (patches inside the level images: true for every keypoint of the extractor, 16 pixels inside its level; OpenCV would throw)
    if (r0 < 0 || r0 + 2 * w + 1 > LG.h || cL0 < 0 || cL0 + 2 * w + 1 > LG.w) return;
    if ((int)(scaleduR0 - Lw - w) < 0 || (int)(scaleduR0 + Lw + w + 1) > LG.w) return;
    const int E = G->edge;
    const uint8_t* bufL = pyr + LG.buf_off;
    const uint8_t* bufR = pyr + (size_t)G->pyr_bytes + LG.buf_off;
    // lanes = pixels of the 11 x 11 patch (two per lane)
    int pl[2], po[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int p = lane + 64 * t;
        const int y = p / 11, x = p - y * 11;
        pl[t] = p < 121 ? (int)bufL[(size_t)(r0 + y + E) * LG.bw + (cL0 + x + E)] : 0;
        po[t] = p < 121 ? (r0 + y + E) * LG.bw + (x + E) : -1;
    }
    int bestD = 0x7fffffff, bestincR = 0;
    float vd[11];
#pragma unroll
    for (int inc = -5; inc <= 5; inc++) {
        const int cR0 = (int)(scaleduR0 + (float)inc - w);
        int s = 0;
#pragma unroll
        for (int t = 0; t < 2; t++) if (po[t] >= 0) s += abs(pl[t] - (int)bufR[(size_t)po[t] + cR0]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        const float dist = (float)s;
        if (dist < (float)bestD) { bestD = (int)dist; bestincR = inc; }
        vd[inc + 5] = dist;
    }
    if (bestincR == -5 || bestincR == 5) return;
    float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
    for (int k = 1; k <= 9; k++) if (k == bestincR + 5) { dist1 = vd[k - 1]; dist2 = vd[k]; dist3 = vd[k + 1]; }
    const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
    if (deltaR < -1 || deltaR > 1) return;
    float bestuR = G->sf[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
    float disparity = uL - bestuR;
    if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = 0.01; bestuR = (float)((double)uL - 0.01); }      // (double literals: :1021-1022)
        if (lane == 0) { depth[iL] = mbf / disparity; uRight[iL] = bestuR; sad[iL] = bestD; }
    }
}

__global__ __launch_bounds__(1024) void stereo_median_kernel(const int32_t* __restrict__ n, int cap, const int32_t* __restrict__ sad,
                                                             float* __restrict__ uRight, float* __restrict__ depth, int32_t* __restrict__ nmatch)
{
    // the norm of rank M / 2 among the M accepted matches ordered by (norm, iL) -- its VALUE does not depend on how equal norms are
    // ordered --: a two-level counting select over the norms' 15 bits (an 11 x 11 patch: <= 30 855)
    __shared__ int s_hist[256];
    __shared__ int s_sel[3];                             // M | chosen high byte, rank inside it | the median
    const int N = min(n[0], cap), tid = threadIdx.x;
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 1024) { const int v = sad[i]; if (v >= 0) atomicAdd(&s_hist[min(v >> 7, 255)], 1); }
    __syncthreads();
    if (tid == 0) {
        int M = 0;
        for (int b = 0; b < 256; b++) M += s_hist[b];
        s_sel[0] = M;
        int k = M / 2, hb = 0;
        while (hb < 255 && k >= s_hist[hb]) { k -= s_hist[hb]; hb++; }
        s_sel[1] = hb; s_sel[2] = k;
        *nmatch = M;
    }
    __syncthreads();
    const int M = s_sel[0], hb = s_sel[1], k = s_sel[2];
    if (M == 0) return;                                  // (the reference reads vDistIdx[0] of an empty vector here)
    __syncthreads();
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 1024) { const int v = sad[i]; if (v >= 0 && min(v >> 7, 255) == hb) atomicAdd(&s_hist[hb == 255 ? min(v - (255 << 7), 255) : (v & 127)], 1); }
    __syncthreads();
    if (tid == 0) {
        int kk = k, lb = 0;
        while (lb < 255 && kk >= s_hist[lb]) { kk -= s_hist[lb]; lb++; }
        s_sel[2] = (hb << 7) + lb;
    }
    __syncthreads();
    const float median = (float)s_sel[2];
    const float thDist = 1.5f * 1.4f * median;
    for (int i = tid; i < N; i += 1024) {
        const int si = sad[i];
        if (si >= 0 && !((float)si < thDist)) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    }
}

int stereo_match_dev(eorb_ctx* c, const eorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, float mb, float mbf,
                     float* d_uright, float* d_depth, int32_t* d_sad, int32_t* d_nmatch)
{
    OrbState& o = c->orb;
    if (!o.configured) return set_err(c, EORB_E_NOTCONF, "stereo matches: eorb_orb_configure not called");
    ProfScope ps(c, "stereo_match");
    const int cap = o.max_out;
    if (cap > 65535) return set_err(c, EORB_E_CAPACITY, "stereo matches: %d keypoints per image exceed the 16-bit candidate index", cap);
    stereo_match_kernel<<<(cap + 3) / 4, 256, 0, c->stream>>>((const DevGeom*)o.geom.p, (const uint8_t*)c->pyr.p, d_kps, d_desc, d_n, cap, mb, mbf,
                                                              d_uright, d_depth, d_sad);
    stereo_median_kernel<<<1, 1024, 0, c->stream>>>(d_n, cap, d_sad, d_uright, d_depth, d_nmatch);
    EORB_LAUNCH_CHECK(c, "stereo match kernels");
    return EORB_OK;
}

// the flag of the last extraction of B slices, copied device-to-device next to the call's other outputs
int orb_err_flag_to(eorb_ctx* c, int B, int32_t* d_dst)
{
    OrbState& o = c->orb;
    const int32_t* err_flag = (const int32_t*)((char*)c->lvl_cnt.p + (size_t)B * o.nlevels * sizeof(int32_t));
    EORB_HIP(c, hipMemcpyAsync(d_dst, err_flag, sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    return EORB_OK;
}

int orb_err_flag(eorb_ctx* c, int B, int* flag)
{
    OrbState& o = c->orb;
    int32_t* err_flag = (int32_t*)((char*)c->lvl_cnt.p + (size_t)B * o.nlevels * sizeof(int32_t));
    EORB_HIP(c, hipMemcpyAsync(flag, err_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, hipStreamSynchronize(c->stream));
    return EORB_OK;
}

}  // namespace eorb
