// ev_accum.hip -- event -> image accumulation on gfx950, bit-exact w.r.t. the sequential CPU loop.
//
// Replaces EvImConverter::ev2im_gauss / ev2im (src/Event/EventConversion.cc:173-269 of the
// reference).  The reference adds one (2h+1)^2 Gaussian stamp per event into a float image
// SEQUENTIALLY; float addition is not associative, so every pixel must receive its contributions in
// event order.  Design (SURVEY App.B H1):
//   K1a ev_count_kernel   entries of every (4096-event chunk, 8x8-pixel tile) by LDS atomics (an event is copied to every
//                        tile its stamp touches).
//   K1b ev_scan_kernel    scan over chunks and tiles: every tile of every slice gets ONE contiguous event-ordered list.
//   K1c ev_scatter_kernel one wavefront per chunk, ORDER-PRESERVING scatter into the lists: lanes = consecutive events,
//                        rank among same-tile lanes by ballot matching; tiles are visited in parity classes so a tile is only
//                        ever targeted in one pass.
//   K2 ev_gather_kernel / K2r ev_gather_raw_kernel   one workgroup per tile, heaviest tiles first: 64-entry batches through a
//                        pipeline of value waves and one add wave (see the comments above the kernels); every pixel adds its
//                        taps in event order.
//                        Running min/max (resolveMinMaxVals :32-39) reduced per wave -> atomics.
//   K3 ev_normalize_kernel  normalizeImage (:67-72): convertTo(CV_8UC1, alpha, beta), round-half-even.
// All kernels are batched over time-slices (blockIdx ranges over slices x tiles / chunks).
#include "eorb_ctx.h"
#include "dev_math.h"
#include "ev_common.h"
#include <math.h>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace eorb {

struct BinParams {
    int W, H, h;          // image size, stamp half window (0 in count mode)
    int TX, TY, NT;       // tiles
    int nbits;            // bits needed for a tile id
    int dup;              // entry capacity per event (R*R)
    int mode_count;       // 1: ev2im (coords rounded, no stamp)
    int pol;
    // raw sensor events (eorb_raw_event) resolved through the undistortion maps: per sensor pixel xi | yi << 16 (int16 each)
    int raw, LW, LH;
    const uint32_t* src_info;
    // hashed != 0: the events are 4-byte records { row of the event's position in src_info / the stamp table | negative polarity << 31 }
    // (0x7fffffff in the low bits: dropped) -- float events in bulk, see dd_insert_kernel
    int hashed;
};
constexpr uint32_t kHashDropped = 0x7fffffffu;

__device__ __forceinline__ bool ev_tile_range(const eorb_event16& e, const BinParams& P, int& tx0, int& tx1, int& ty0, int& ty1)
{
    const float x = e.x, y = e.y;
    if (!(x == x && y == y)) return false;          // NaN coordinates (e.g. a one-event MCI window: 0 * inf) convert to
                                                    // INT_MIN on the reference's x86: never in the image
    const int xi = P.mode_count ? (int)roundf(x) : (int)floorf(x);
    const int yi = P.mode_count ? (int)roundf(y) : (int)floorf(y);
    tx0 = max((xi - P.h) >> 3, 0); tx1 = min((xi + P.h) >> 3, P.TX - 1);
    ty0 = max((yi - P.h) >> 3, 0); ty1 = min((yi + P.h) >> 3, P.TY - 1);
    return true;
}

// raw event -> sensor pixel index and its integer image position; events off the maps or (checkInImage) mapped outside the
// image carry xi = yi = -32768 and therefore touch no tile
__device__ __forceinline__ bool ev_tile_range_raw(const eorb_raw_event& q, const BinParams& P, int& tx0, int& tx1, int& ty0, int& ty1,
                                                  uint32_t& src, uint32_t& info)
{
    if ((int)q.x >= P.LW || (int)q.y >= P.LH) return false;
    src = (uint32_t)q.y * (uint32_t)P.LW + q.x;
    info = P.src_info[src];
    const int xi = (int)(int16_t)(info & 0xffff), yi = (int)(int16_t)(info >> 16);
    tx0 = max((xi - P.h) >> 3, 0); tx1 = min((xi + P.h) >> 3, P.TX - 1);
    ty0 = max((yi - P.h) >> 3, 0); ty1 = min((yi + P.h) >> 3, P.TY - 1);
    return true;
}

// K1a: entries of every (chunk, tile)
__global__ __launch_bounds__(256) void ev_count_kernel(const eorb_event16* __restrict__ ev, const ChunkDesc* __restrict__ chunks,
                                                       BinParams P, uint16_t* __restrict__ segcnt)
{
    extern __shared__ uint32_t cnt[];               // NT
    const ChunkDesc cd = chunks[blockIdx.x];
    const int NT = P.NT;
    for (int i = threadIdx.x; i < NT; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const eorb_event16* e = ev + cd.start;
    if (P.raw) {
        // eight events per thread and round: all event loads, then all map lookups, then the counting (the chain event -> map entry
        // -> tile range is latency; the loads of the round are in flight together)
        constexpr int U = 8;
        for (int k0 = threadIdx.x; k0 < cd.n; k0 += blockDim.x * U) {
            uint32_t xy[U]; uint32_t info[U];
            if (P.hashed) {
                const uint32_t* e4 = (const uint32_t*)ev + cd.start;
                uint32_t id[U];
#pragma unroll
                for (int u = 0; u < U; u++) { const int k = k0 + u * blockDim.x; xy[u] = k < cd.n ? e4[k] : kHashDropped; }
#pragma unroll
                for (int u = 0; u < U; u++) id[u] = (xy[u] & kHashDropped) != kHashDropped ? (xy[u] & kHashDropped) : 0xffffffffu;
#pragma unroll
                for (int u = 0; u < U; u++) info[u] = id[u] != 0xffffffffu ? P.src_info[id[u]] : 0x80008000u;
            } else {
#pragma unroll
            for (int u = 0; u < U; u++) { const int k = k0 + u * blockDim.x; xy[u] = k < cd.n ? *(const uint32_t*)&((const eorb_raw_event*)e)[k] : 0xffffffffu; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int x = (int)(xy[u] & 0xffff), y = (int)(xy[u] >> 16);
                const bool ok = x < P.LW && y < P.LH;
                info[u] = ok ? P.src_info[(uint32_t)y * (uint32_t)P.LW + x] : 0x80008000u;      // (-32768, -32768): touches no tile
            }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int xi = (int)(int16_t)(info[u] & 0xffff), yi = (int)(int16_t)(info[u] >> 16);
                const int tx0 = max((xi - P.h) >> 3, 0), tx1 = min((xi + P.h) >> 3, P.TX - 1);
                const int ty0 = max((yi - P.h) >> 3, 0), ty1 = min((yi + P.h) >> 3, P.TY - 1);
                for (int ty = ty0; ty <= ty1; ty++)
                    for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&cnt[ty * P.TX + tx], 1u);
            }
        }
    } else {
        for (int k = threadIdx.x; k < cd.n; k += blockDim.x) {
            int tx0, tx1, ty0, ty1;
            if (ev_tile_range(e[k], P, tx0, tx1, ty0, ty1))
                for (int ty = ty0; ty <= ty1; ty++)
                    for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&cnt[ty * P.TX + tx], 1u);
        }
    }
    __syncthreads();
    uint16_t* sc = segcnt + (size_t)blockIdx.x * NT;
    for (int i = threadIdx.x; i < NT; i += blockDim.x) sc[i] = (uint16_t)cnt[i];
}

// K1b: one workgroup per slice.  Per tile: exclusive scan of its counts over the slice's chunks (segbase) and the total
// (tile_cnt); then an exclusive scan of the totals over the tiles (tile_base): every tile's entries become ONE contiguous,
// event-ordered list, so K2 runs full 64-entry batches whatever the chunking.
__global__ __launch_bounds__(1024) void ev_scan_kernel(const int* __restrict__ slice_chunk0, const uint16_t* __restrict__ segcnt, int NT,
                                                       uint32_t* __restrict__ segbase, uint32_t* __restrict__ tile_cnt,
                                                       uint32_t* __restrict__ tile_base)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int slice = blockIdx.x;
    const int c0 = slice_chunk0[slice], c1 = slice_chunk0[slice + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < NT; t0 += blockDim.x) {
        const int tile = t0 + threadIdx.x;
        uint32_t run = 0;
        if (tile < NT) {
            int c = c0;
            for (; c + 8 <= c1; c += 8) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = segcnt[(size_t)(c + u) * NT + tile];
#pragma unroll
                for (int u = 0; u < 8; u++) { segbase[(size_t)(c + u) * NT + tile] = run; run += v[u]; }
            }
            for (; c < c1; c++) { const uint32_t v = segcnt[(size_t)c * NT + tile]; segbase[(size_t)c * NT + tile] = run; run += v; }
            tile_cnt[(size_t)slice * NT + tile] = run;
        }
        uint32_t incl = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (tile < NT) tile_base[(size_t)slice * NT + tile] = before + incl - run;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry = before + incl;
        __syncthreads();
    }
}

// K1c: order-preserving scatter.  One wavefront per chunk: lanes = consecutive events, rank among the lanes that target the
// same tile by ballot matching; tiles are visited in parity classes so a tile is only ever targeted in one pass.
template <int R, bool POL>
__global__ __launch_bounds__(64) void ev_scatter_kernel(const eorb_event16* __restrict__ ev, const ChunkDesc* __restrict__ chunks,
                                                        BinParams P, const int64_t* __restrict__ slice_ebase,
                                                        const uint32_t* __restrict__ segbase, const uint32_t* __restrict__ tile_base,
                                                        float* __restrict__ entries)
{
    extern __shared__ uint32_t cnt[];               // NT: write cursor of every tile list
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    const ChunkDesc cd = chunks[chunk];
    const int NT = P.NT;
    for (int i = lane; i < NT; i += 64) cnt[i] = tile_base[(size_t)cd.slice * NT + i] + segbase[(size_t)chunk * NT + i];
    __syncthreads();
    const eorb_event16* e = ev + cd.start;
    constexpr int ESZ = POL ? 4 : 2;
    float* out = entries + (size_t)slice_ebase[cd.slice] * ESZ;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int s = 0; s < cd.n; s += 64) {
        const int k = s + lane;
        bool valid = k < cd.n;
        float x = 0.f, y = 0.f, sg = 1.f;
        int tx0 = 1, tx1 = 0, ty0 = 1, ty1 = 0;
        if (valid) {
            if (P.raw) {
                // entry = { sensor pixel | negative polarity << 31, xi | yi << 16 } in the float2 slot
                const eorb_raw_event q = ((const eorb_raw_event*)e)[k];
                uint32_t src = 0, info = 0;
                valid = ev_tile_range_raw(q, P, tx0, tx1, ty0, ty1, src, info);
                x = __uint_as_float(src | (q.p ? 0u : 0x80000000u)); y = __uint_as_float(info);
            } else {
                const eorb_event16 q = e[k];
                valid = ev_tile_range(q, P, tx0, tx1, ty0, ty1);
                x = q.x; y = q.y;
                if (POL) sg = (__double_as_longlong(q.t) < 0) ? -1.0f : 1.0f;
            }
        }
#pragma unroll
        for (int cy = 0; cy < R; cy++) {
#pragma unroll
            for (int cx = 0; cx < R; cx++) {
                // the tile of residue class (cx, cy) inside this event's tile range, if any
                const int tx = tx0 + ((cx - tx0 % R) + R) % R;
                const int ty = ty0 + ((cy - ty0 % R) + R) % R;
                const bool has = valid && tx <= tx1 && ty <= ty1;
                uint64_t m = __ballot(has);
                if (m == 0ull) continue;
                const int key = has ? ty * P.TX + tx : 0;
                for (int b = 0; b < P.nbits; b++) {
                    const bool bit = (key >> b) & 1;
                    const uint64_t bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                if (has) {
                    const int rank = __popcll(m & lt_mask);
                    const uint32_t base = cnt[key];
                    const size_t pos = (size_t)base + rank;
                    if (POL) { float4 v = make_float4(x, y, sg, 0.f); *(float4*)(out + pos * 4) = v; }
                    else { float2 v = make_float2(x, y); *(float2*)(out + pos * 2) = v; }
                    if (rank == 0) cnt[key] = base + (uint32_t)__popcll(m);
                }
                __syncthreads();
            }
        }
    }
}

// K1c, second form: the chunk's entries are SORTED BY TILE IN LDS first and leave the CU as contiguous runs (one run per tile: the
// chunk's segment of that tile's list), so that a store instruction covers a few cache lines instead of up to 64 (the first form
// is bound by the rate of its scattered 8-byte store requests, ~100 G/s).  One 4-wave workgroup per chunk (<= 2048 events); wave w
// owns the w-th quarter of the chunk's events, whose tile ranges it keeps in registers.
//   A  every wave counts its quarter's entries per tile (LDS atomics, its own row of cntw).
//   B  per tile: exclusive prefix of the four rows (wave w's entries of a tile come after those of the waves before it) and an
//      exclusive scan of the totals over the tiles: loff[t] = start of tile t's run inside the chunk's sorted order.
//   C  every wave walks its sub-batches of 64 events in order; rank among the lanes of the same tile by ballot matching (tiles
//      visited in parity classes, as in the first form) + the wave's running counter of that tile: sidx[slot] = event.  The
//      counters are private to the wave: no barrier inside this phase.
//   D  slot p of the sorted order -> (event, tile) -> the entry is rebuilt from the event (L2) and stored at the tile's run base +
//      (p - loff[tile]): consecutive threads write consecutive entries of a run.
constexpr int kScatWaves = 8;
template <int R>
__global__ __launch_bounds__(64 * kScatWaves) void ev_scatter2_kernel(const eorb_event16* __restrict__ ev, const ChunkDesc* __restrict__ chunks,
                                                          BinParams P, int chunk_cap, const int64_t* __restrict__ slice_ebase,
                                                          const uint32_t* __restrict__ segbase, const uint32_t* __restrict__ tile_base,
                                                          uint2* __restrict__ entries)
{
    extern __shared__ unsigned char sm2[];
    __shared__ uint32_t s_wsum[kScatWaves];
    constexpr int NTHR = 64 * kScatWaves;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x;
    const ChunkDesc cd = chunks[chunk];
    const int NT = P.NT, NTp = (NT + 1) & ~1;
    uint2* pay = (uint2*)sm2;                                         // chunk_cap: the 8-byte entry of every event
    uint16_t* prng = (uint16_t*)(pay + chunk_cap);                    // chunk_cap: first tile of the event's range, tx0 | ty0 << 8
    uint16_t* sidx = prng + chunk_cap;                                // chunk_cap * R * R: slot of the sorted order -> event | dx << 11 | dy << 13
    uint16_t* cntw = sidx + (size_t)chunk_cap * R * R;                // kScatWaves * NTp
    uint16_t* loff = cntw + kScatWaves * NTp;                         // NT + 1 (+ 1 pad)
    uint32_t* gbase = (uint32_t*)(loff + NTp + 2);                    // NT: first entry of the tile's run in the global lists
    for (int i = tid; i < kScatWaves * NTp / 2; i += NTHR) ((uint32_t*)cntw)[i] = 0u;
    const eorb_event16* e = ev + cd.start;
    // share of wave w: events [w * Q, (w + 1) * Q), Q a multiple of 64; S sub-batches of 64 (<= 4: chunk_cap <= 2048)
    const int Q = (((cd.n + kScatWaves - 1) / kScatWaves) + 63) & ~63;
    const int S = Q >> 6;
    constexpr int SMAX = 4;
    uint32_t rng[SMAX];                                               // tx0 | ty0 << 8 | (tx1 - tx0 + 1) << 16 | (ty1 - ty0 + 1) << 20; 0 = no entry
    __syncthreads();
    // ---- A: entries into LDS, tile ranges into registers, counts per (wave, tile) ----
    uint32_t* cw32 = (uint32_t*)(cntw + wave * NTp);
    // (raw events: the event loads of all sub-batches first, then the map lookups, then the LDS work -- the chain event -> map
    // entry -> range is latency, so the loads of the share are in flight together)
    uint32_t rsrc[SMAX], rneg[SMAX], rinfo[SMAX];                   // table row of the event (0xffffffff: none), negative polarity, map entry
    if (P.raw) {
        if (P.hashed) {
            const uint32_t* e4 = (const uint32_t*)ev + cd.start;
#pragma unroll
            for (int s = 0; s < SMAX; s++) {
                const int k = wave * Q + s * 64 + lane;
                const uint32_t rec = (s < S && k < cd.n) ? e4[k] : kHashDropped;
                rneg[s] = rec >> 31; rsrc[s] = rec & kHashDropped;
            }
#pragma unroll
            for (int s = 0; s < SMAX; s++) rsrc[s] = rsrc[s] != kHashDropped ? rsrc[s] : 0xffffffffu;
        } else {
#pragma unroll
            for (int s = 0; s < SMAX; s++) {
                const int k = wave * Q + s * 64 + lane;
                const uint2 q = (s < S && k < cd.n) ? *(const uint2*)&((const eorb_raw_event*)e)[k] : make_uint2(0xffffffffu, 0u);
                const int x = (int)(q.x & 0xffff), y = (int)(q.x >> 16);
                rsrc[s] = (x < P.LW && y < P.LH) ? (uint32_t)y * (uint32_t)P.LW + x : 0xffffffffu;
                rneg[s] = q.y ? 0u : 1u;
            }
        }
#pragma unroll
        for (int s = 0; s < SMAX; s++) rinfo[s] = rsrc[s] != 0xffffffffu ? P.src_info[rsrc[s]] : 0u;
    }
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        rng[s] = 0u;
        const int k = wave * Q + s * 64 + lane;
        if (s < S && k < cd.n) {
            int tx0 = 1, tx1 = 0, ty0 = 1, ty1 = 0;
            bool ok;
            uint2 pl;
            if (P.raw) {
                // entry = { sensor pixel | negative polarity << 31, xi | yi << 16 }
                ok = rsrc[s] != 0xffffffffu;
                const uint32_t src = ok ? rsrc[s] : 0u, info = rinfo[s];
                if (ok) {
                    const int xi = (int)(int16_t)(info & 0xffff), yi = (int)(int16_t)(info >> 16);
                    tx0 = max((xi - P.h) >> 3, 0); tx1 = min((xi + P.h) >> 3, P.TX - 1);
                    ty0 = max((yi - P.h) >> 3, 0); ty1 = min((yi + P.h) >> 3, P.TY - 1);
                }
                pl = make_uint2(src | (rneg[s] ? 0x80000000u : 0u), info);
            } else {
                const eorb_event16 q = e[k];
                ok = ev_tile_range(q, P, tx0, tx1, ty0, ty1);
                pl = make_uint2(__float_as_uint(q.x), __float_as_uint(q.y));
            }
            pay[k] = pl;
            prng[k] = (uint16_t)((tx0 & 0xff) | ((ty0 & 0xff) << 8));
            if (ok && tx1 >= tx0 && ty1 >= ty0) {
                rng[s] = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)(tx1 - tx0 + 1) << 16) | ((uint32_t)(ty1 - ty0 + 1) << 20);
                for (int ty = ty0; ty <= ty1; ty++)
                    for (int tx = tx0; tx <= tx1; tx++) {
                        const int t = ty * P.TX + tx;
                        atomicAdd(&cw32[t >> 1], 1u << (16 * (t & 1)));       // 16-bit counters, two per word (a share has <= 256 events)
                    }
            }
        }
    }
    __syncthreads();
    // ---- B: per tile the exclusive prefix over the waves; exclusive scan of the totals over the tiles ----
    {
        const int per = (NT + NTHR - 1) / NTHR;                       // consecutive tiles of one thread
        const int t0 = tid * per, t1 = min(t0 + per, NT);
        uint32_t mine = 0;
        for (int t = t0; t < t1; t++) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < kScatWaves; w++) { const uint32_t v = cntw[w * NTp + t]; cntw[w * NTp + t] = (uint16_t)run; run += v; }
            loff[t] = (uint16_t)run;                                  // total of the tile, turned into its offset below
            mine += run;
            gbase[t] = tile_base[(size_t)cd.slice * NT + t] + segbase[(size_t)chunk * NT + t];
        }
        uint32_t incl = (uint32_t)wave_incl_scan((int)mine);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w = 0; w < wave; w++) before += s_wsum[w];
        for (int t = t0; t < t1; t++) { const uint32_t v = loff[t]; loff[t] = (uint16_t)before; before += v; }
        if (tid == NTHR - 1) loff[NT] = (uint16_t)before;             // (threads past the last tile carry the grand total)
    }
    __syncthreads();
    // ---- C: stable ranks -> sidx ----
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint16_t* cw = cntw + wave * NTp;
    const int txr = (P.TX + R - 1) / R;
    int mbits = 1; while ((1 << mbits) < txr * ((P.TY + R - 1) / R)) mbits++;
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        if (s >= S) break;
        const uint32_t rg = rng[s];
        const int tx0 = rg & 0xff, ty0 = (rg >> 8) & 0xff, tx1 = tx0 + (int)((rg >> 16) & 0xf) - 1, ty1 = ty0 + (int)((rg >> 20) & 0xf) - 1;
        const bool valid = rg != 0u;
        const uint16_t kloc = (uint16_t)(wave * Q + s * 64 + lane);
#pragma unroll
        for (int cy = 0; cy < R; cy++) {
#pragma unroll
            for (int cx = 0; cx < R; cx++) {
                // the tile of residue class (cx, cy) inside this event's tile range, if any
                const int tx = tx0 + ((cx - tx0 % R) + R) % R;
                const int ty = ty0 + ((cy - ty0 % R) + R) % R;
                const bool has = valid && tx <= tx1 && ty <= ty1;
                uint64_t m = __ballot(has);
                if (m == 0ull) continue;
                const int key = has ? ty * P.TX + tx : 0;
                // lanes of the same tile: inside one parity class a tile is identified by (tx / R, ty / R) -- fewer bits to match
                const int mkey = has ? (ty / R) * txr + (tx / R) : 0;
                for (int b = 0; b < mbits; b++) {
                    const bool bit = (mkey >> b) & 1;
                    const uint64_t bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                if (has) {
                    const int rank = __popcll(m & lt_mask);
                    const uint32_t base = cw[key];
                    sidx[(uint32_t)loff[key] + base + rank] = (uint16_t)(kloc | ((tx - tx0) << 11) | ((ty - ty0) << 13));
                    if (rank == 0) cw[key] = (uint16_t)(base + (uint32_t)__popcll(m));
                }
            }
        }
    }
    __syncthreads();
    // ---- D: the sorted order leaves as contiguous runs: consecutive threads write consecutive entries of a tile's run ----
    uint2* out = entries + (size_t)slice_ebase[cd.slice];
    const int E = loff[NT];
    for (int p = tid; p < E; p += NTHR) {
        const uint32_t sv = sidx[p];
        const int k = sv & 0x7ff;
        const uint32_t r0 = prng[k];
        const int t = ((int)(r0 >> 8) + (int)((sv >> 13) & 3)) * P.TX + (int)(r0 & 0xff) + (int)((sv >> 11) & 3);
        const uint2 v = pay[k];
        const size_t pos = (size_t)gbase[t] + (uint32_t)(p - (int)loff[t]);
        out[pos] = v;
    }
}

// Heaviest-first launch order for K2 (longest-processing-time-first): event data is spatially concentrated (on the reference's
// `shapes` sequences one tile holds ~6 % of a slice's entries), and a workgroup that starts its long tile late is the tail of the
// launch.  Bucket sort of the (slice, tile) work items by a 2-bits-per-octave log weight.
__device__ __forceinline__ int ev_weight_bucket(uint32_t w)
{
    if (w == 0) return 63;
    const int lg = 31 - __clz(w);                                  // floor(log2 w)
    const int half = (lg > 0) ? (int)((w >> (lg - 1)) & 1u) : 0;
    return 62 - min(62, 2 * lg + half);                            // 0 = heaviest
}
constexpr int kOrderItems = 4096;       // work items per workgroup of the two kernels below
// (1) bucket histogram of every block of kOrderItems work items
__global__ __launch_bounds__(1024) void ev_tile_hist_kernel(const uint32_t* __restrict__ weight, int total, uint32_t* __restrict__ blk_hist)
{
    __shared__ uint32_t hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const int i0 = blockIdx.x * kOrderItems, i1 = min(i0 + kOrderItems, total);
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) atomicAdd(&hist[ev_weight_bucket(weight[i])], 1u);
    __syncthreads();
    if (threadIdx.x < 64) blk_hist[blockIdx.x * 64 + threadIdx.x] = hist[threadIdx.x];
}
// (2) a block's items of bucket b go behind all heavier buckets and behind the bucket-b items of the blocks before it
__global__ __launch_bounds__(1024) void ev_tile_order_kernel(const uint32_t* __restrict__ weight, int total, const uint32_t* __restrict__ blk_hist,
                                                             int32_t* __restrict__ order)
{
    __shared__ uint32_t tot[64], before[64], cur[64];
    if (threadIdx.x < 64) tot[threadIdx.x] = before[threadIdx.x] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < (int)gridDim.x * 64; k += blockDim.x) {
        const uint32_t v = blk_hist[k];
        if (!v) continue;
        atomicAdd(&tot[k & 63], v);
        if ((k >> 6) < (int)blockIdx.x) atomicAdd(&before[k & 63], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 64; b++) { cur[b] = run + before[b]; run += tot[b]; }
    }
    __syncthreads();
    const int i0 = blockIdx.x * kOrderItems, i1 = min(i0 + kOrderItems, total);
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) order[atomicAdd(&cur[ev_weight_bucket(weight[i])], 1u)] = i;
}

struct GatherParams {
    int W, H, h, TX, TY, NT;
    int mode_count;
    int total;           // number of (slice, tile) work items
    float two_sig2;      // 2.0f * sig2
    float norm;          // 2.0f*float(CV_PI)*sig2
    float inv_two_sig2;  // exact reciprocal when two_sig2 is a power of two
    float rcp_norm;      // RN(1/norm)
    int div_is_pow2, fast_norm;
    const float* stamps;  // RAW: per sensor pixel the (2h+1)^2 stamp values, [column][row]
    int stamp_stride, stamp_colstride;   // floats per sensor pixel / per (16-byte padded) stamp column
};

// K2: the gather for float events (values from expf) and for count images; raw sensor events with a Gaussian stamp take K2r below.
// One 512-thread workgroup per 8x8 tile, heaviest tiles first (EORB_GATHER_THREADS overrides: 384..704 measured, 512 fastest;
// 4 workgroups per CU: 8 waves/SIMD at <= 64 VGPRs, 39 KB of LDS).  The tile's entries are consumed in batches of 64 (event
// order) through a 3-stage software pipeline with ONE barrier per batch:
//   wave 1, set-up(t)   lane = entry (loaded one batch ahead): integer position / residuals (breakFloatCoords :51-57) and the
//                       tile-local rectangle of stamp taps.  A DPP prefix sum of the rectangle widths lays the entries' stamp
//                       columns side by side; owner[column] = entry.
//   waves 2-7, values(t-1)  lane = one stamp column (a round = 64 columns): its rows' values -- exp_XY2f (:59-65) evaluated in
//                       f64 like glibc's expf -- go to slot e (the entry's position in the batch) of the pixels' lists:
//                       vals[pixel][e].
//   wave 0, adds(t-2)   lane = pixel: acc += list[e], e = 0..63 (newVal = image + polSign*val, :251-254): the only sequential
//                       part.  Slots of entries that do not touch the pixel hold +0.0f, and x + 0.0f == x bit for bit, so the
//                       order of the real adds is the event order and no masks, counts or ranks are needed; lists are read four
//                       slots at a time (ds_read_b128) and cleared behind the read.
// With pol == false every increment is >= 0, so the running max is the final value and the running min stays 0.
struct EvEntryInfo { uint32_t xy; float xr, yr, sg; };      // xi | yi << 16 (int16 each)
constexpr int kValStride = 68;     // floats per pixel list: 64 ranks, padded so 16 lanes' ds_read_b128 hit 16 distinct bank groups
#ifndef EORB_GATHER_THREADS
#define EORB_GATHER_THREADS 512
#endif
constexpr int kGatherThreads = EORB_GATHER_THREADS;

#ifdef EORB_DIAG
__device__ unsigned long long g_diag[16];
#endif
#ifdef EORB_TRACE
// timeline of the heaviest tile's workgroup: [wave][batch - 200][stamp], shader-clock ticks
__device__ unsigned long long g_trace[16 * 64 * 8];
#define EORB_TR(k) do { if (blockIdx.x == 0 && t >= 200 && t < 264 && lane == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    g_trace[(wave * 64 + (t - 200)) * 8 + (k)] = __builtin_readcyclecounter(); } } while (0)
#else
#define EORB_TR(k) do { } while (0)
#endif

// MODE 0: general sigma (IEEE divisions); 1: 2*sig2 a power of two and reciprocal+fma normalisation (sigma = 1, 0.5, 2 ...);
// 2: count image (ev2im)
// 8 waves/SIMD (<= 64 VGPRs, no spills): four 512-thread workgroups per CU instead of three; +7..13 % measured
#ifndef EORB_GATHER_WPE
#define EORB_GATHER_WPE 8
#endif
#define EORB_GATHER_ATTR __attribute__((amdgpu_waves_per_eu(EORB_GATHER_WPE, EORB_GATHER_WPE)))
#ifndef EORB_GATHER_U
#define EORB_GATHER_U 2
#endif
template <bool POL, int MODE, bool RAW>
__global__ __launch_bounds__(1024) EORB_GATHER_ATTR void ev_gather_kernel(const int64_t* __restrict__ slice_ebase, // B: first entry of the slice
                                                                  const int32_t* __restrict__ order,      // work items, heaviest first
                                                                  GatherParams P, const uint32_t* __restrict__ tile_cnt,
                                                                  const uint32_t* __restrict__ tile_base,
                                                                  const float* __restrict__ entries,
                                                                  float* __restrict__ img, uint32_t* __restrict__ minmax_enc)
{
    __shared__ uint64_t tab[32];
    __shared__ EvEntryInfo einfo[2][64];
    __shared__ uint32_t rinfo[2][64];               // a0 | b0 << 4 | w << 8 | h << 12 (tile-local tap rectangle) | first column << 16
    __shared__ uint8_t owner[2][512];               // stamp column (entries' rectangles laid side by side) -> entry
    __shared__ int ncols[2];
    __shared__ int any_ok;                          // some entry of the tile touches an in-image pixel
    __shared__ __attribute__((aligned(16))) float vals[2][kValStride * 64 + 64];  // [pixel][rank] (+64: sink rows for masked stores)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = (int)(blockDim.x >> 6), nprod = nwaves - 2;      // wave 0 adds, wave 1 set-up, the rest values
    if (tid < 32) tab[tid] = kExp2Tab[tid];
    if (tid == 0) any_ok = 0;
    for (int i = tid; i < 2 * (kValStride * 64 + 64); i += blockDim.x) (&vals[0][0])[i] = 0.f;
    const int logical = order[blockIdx.x];
    const int slice = logical / P.NT;
    const int tile = logical - slice * P.NT;
    const int tx0 = (tile % P.TX) * kTile, ty0 = (tile / P.TX) * kTile;
    const int lx = lane & 7, ly = lane >> 3;
    const int px = tx0 + lx, py = ty0 + ly;
    const bool inimg = px < P.W && py < P.H;
    const int xhi = min(tx0 + kTile - 1, P.W - 1), yhi = min(ty0 + kTile - 1, P.H - 1);
    constexpr int ESZ = RAW ? 2 : (POL ? 4 : 2);
    const int h = P.h;
    // the tile constants the set-up wave uses per entry live in VGPRs: as scalars they are spilled (the kernel needs > 102 SGPRs)
    // and reloaded with v_readlane in every batch
    int v_tx0 = tx0, v_ty0 = ty0, v_xhi = xhi, v_yhi = yhi, v_h = h;
    asm volatile("" : "+v"(v_tx0), "+v"(v_ty0), "+v"(v_xhi), "+v"(v_yhi), "+v"(v_h));       // measured: 5.15 -> 4.95 ms at B=64
    __syncthreads();
    // the tile's entries: one contiguous event-ordered list (K1b/K1c)
    const int nent = (int)tile_cnt[logical];
    const int nbatch = (nent + 63) >> 6;
    const float* list = entries + ((size_t)slice_ebase[slice] + tile_base[logical]) * ESZ;
    float acc = 0.0f, vmax = -1000000.0f, vmin = 0.0f;
    // wave 1 keeps the next batch in registers (loaded one iteration ahead)
    int jnext = 0;
    float ex = 0.f, ey = 0.f, esg = 1.f; bool valid = false;
    auto load_batch = [&]() {
        const int j = jnext + lane;
        valid = j < nent;
        if (valid) {
            if (!RAW && POL) { float4 v = *(const float4*)(list + (size_t)j * 4); ex = v.x; ey = v.y; esg = v.z; }
            else { float2 v = *(const float2*)(list + (size_t)j * 2); ex = v.x; ey = v.y; }
        }
        jnext += 64;
    };
    if (wave == 1 && nbatch > 0) load_batch();
#ifdef EORB_GATHER_PRIO
    if (wave <= 1) __builtin_amdgcn_s_setprio(3);          // the serial stages win issue arbitration over the value waves
#endif
#ifdef EORB_DIAG
    unsigned long long d_work = 0, d_t0 = __builtin_readcyclecounter(), d_setup = 0;
#endif
    for (int t = 0; t < nbatch + 2; t++) {
#ifdef EORB_DIAG
        const unsigned long long d_s = __builtin_readcyclecounter();
#endif
        EORB_TR(0);
        if (wave == 0) {
            // ---- adds(t-2): slot e of a pixel's list holds entry e's tap, or +0.0f when that entry does not touch the pixel.
            //      Adding +0.0f leaves acc unchanged bit for bit (acc is never -0.0), so the wave adds all 64 slots in order
            //      without masks, counts or ranks, and clears the list behind the read ----
            if (t >= 2) {
                const int bs2 = t & 1;
                EORB_TR(1);
                float4* vb = (float4*)(vals[bs2] + lane * kValStride);
                const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#ifndef EORB_KO_ADDS
#pragma unroll
                for (int g = 0; g < 16; g += 4) {
                    const float4 v0 = vb[g], v1 = vb[g + 1], v2 = vb[g + 2], v3 = vb[g + 3];
                    vb[g] = zero4; vb[g + 1] = zero4; vb[g + 2] = zero4; vb[g + 3] = zero4;
                    if (!POL) {
                        acc = acc + v0.x; acc = acc + v0.y; acc = acc + v0.z; acc = acc + v0.w;
                        acc = acc + v1.x; acc = acc + v1.y; acc = acc + v1.z; acc = acc + v1.w;
                        acc = acc + v2.x; acc = acc + v2.y; acc = acc + v2.z; acc = acc + v2.w;
                        acc = acc + v3.x; acc = acc + v3.y; acc = acc + v3.z; acc = acc + v3.w;
                    } else {
                        // running extremes (resolveMinMaxVals :32-39) only move on a real add: a tap is never 0
#define EORB_ADD(val) { if ((val) != 0.0f) { acc = acc + (val); vmax = fmaxf(vmax, acc); vmin = fminf(vmin, acc); } }
                        EORB_ADD(v0.x) EORB_ADD(v0.y) EORB_ADD(v0.z) EORB_ADD(v0.w) EORB_ADD(v1.x) EORB_ADD(v1.y) EORB_ADD(v1.z) EORB_ADD(v1.w)
                        EORB_ADD(v2.x) EORB_ADD(v2.y) EORB_ADD(v2.z) EORB_ADD(v2.w) EORB_ADD(v3.x) EORB_ADD(v3.y) EORB_ADD(v3.z) EORB_ADD(v3.w)
#undef EORB_ADD
                    }
                }
#endif
            }
        } else {
            if (wave == 1 && t < nbatch) {
                // ---- set-up(t) from the registers loaded one iteration ago ----
                const int bs2 = t & 1;
                // straight-line decode: lanes without an entry get a position far outside the image, which makes the rectangle empty
                int xi, yi; float xr = 0.f, yr = 0.f;
                if (RAW) {                                                               // integer position from the maps (K1c)
                    const uint32_t w0 = __float_as_uint(ex), w1 = __float_as_uint(ey);
                    xi = (int)(int16_t)(w1 & 0xffff); yi = (int)(int16_t)(w1 >> 16);
                    esg = (w0 >> 31) ? -1.0f : 1.0f;
                } else if (MODE == 2) { xi = (int)roundf(ex); yi = (int)roundf(ey); }   // roundFloatCoord :46-49
                else {                                                                   // breakFloatCoords :51-57
                    xi = (int)floorf(ex); yi = (int)floorf(ey);
                    xr = ex - (float)xi; yr = ey - (float)yi;
                }
                xi = valid ? xi : -32768;
                const int a0 = max(xi - v_h, v_tx0) - v_tx0, a1 = min(xi + v_h, v_xhi) - v_tx0;
                const int b0 = max(yi - v_h, v_ty0) - v_ty0, b1 = min(yi + v_h, v_yhi) - v_ty0;
                const bool ok = a1 >= a0 && b1 >= b0;
                const int ra0 = a0 & 15, rb0 = b0 & 15, rw = ok ? a1 - a0 + 1 : 0, rh = ok ? b1 - b0 + 1 : 0;
                EORB_TR(1);
                EvEntryInfo ei; ei.xy = (uint32_t)(xi & 0xffff) | ((uint32_t)(yi & 0xffff) << 16); ei.xr = xr; ei.yr = yr; ei.sg = esg;
                einfo[bs2][lane] = ei;
                const int incl = wave_incl_scan(rw);
                const int coff = incl - rw;
                rinfo[bs2][lane] = (uint32_t)ra0 | ((uint32_t)rb0 << 4) | ((uint32_t)rw << 8) | ((uint32_t)rh << 12) | ((uint32_t)coff << 16);
#pragma unroll
                for (int k = 0; k < 8; k++) if (k < rw) owner[bs2][coff + k] = (uint8_t)lane;
                if (lane == 63) ncols[bs2] = incl;
                if (__any(ok) && lane == 0) any_ok = 1;
                EORB_TR(2);
                EORB_TR(3);
                if (t + 1 < nbatch) load_batch();       // prefetch batch t+1
#ifdef EORB_DIAG
                d_setup += __builtin_readcyclecounter() - d_s;
#endif
            }
            // ---- values(t-1): lane = one stamp column of one entry (the set-up wave laid the entries' tile-local rectangles
            //      side by side: column g belongs to entry owner[g]); the lane walks the column's rows, two at a time ----
            if (wave >= 2 && t >= 1 && t <= nbatch) {
                const int bs2 = (t - 1) & 1;
                const int C = ncols[bs2];
                EORB_TR(1);
                float* vbase = vals[bs2];
                float* sink = vbase + kValStride * 64;                  // masked-off rows store here (never read)
#ifdef EORB_KO_VALS
                if (false)
#endif
                for (int g0 = (wave - 2) * 64; g0 < C; g0 += nprod * 64) {
                    const int g = g0 + lane;
                    const bool act = g < C;
                    const int e = act ? (int)owner[bs2][g] : 0;
                    const uint32_t ri = rinfo[bs2][e];
                    const EvEntryInfo ei = einfo[bs2][e];
                    const int qx = (int)(ri & 15u) + (g - (int)(ri >> 16));
                    const int b0 = (int)((ri >> 4) & 15u);
                    const int rh = act ? (int)((ri >> 12) & 15u) : 0;
                    const int xi = (int)(int16_t)(ei.xy & 0xffff), yi = (int)(int16_t)(ei.xy >> 16);
                    if (g0 == (wave - 2) * 64) EORB_TR(2); else EORB_TR(4);
                    const int dy0 = ty0 + b0 - yi;
                    const int pix0 = act ? b0 * 8 + qx : 0;              // pixel of the column's first row; rows step by 8
                    float* vcol = vbase + __umul24(pix0, kValStride) + e; // the entry's slot in that pixel's list
                    const float fx = (float)(tx0 + qx - xi) - ei.xr;            // exp_XY2f(i-xRes, j-yRes) :59-65
                    const float xx = fx * fx;
                    constexpr int U = EORB_GATHER_U;
                    for (int jj = 0; __any(jj < rh); jj += U) {
                        bool on[U]; float v[U];
#pragma unroll
                        for (int u = 0; u < U; u++) on[u] = jj + u < rh;
                        if (MODE == 1) {
                            // four-stage form of glibc's expf so the dependent f64 chains of the U rows interleave.
                            // dd /= 2*sig2 with a power-of-two divisor == product with its exact reciprocal (same real number,
                            // same rounding); ev / norm = correctly rounded quotient from the correctly rounded reciprocal
                            // (Markstein), valid while the residual is a normal float: the host selects MODE 1 only when
                            // exp(-dd) > 1e-27 for every tap; proven against IEEE division by tests/test_gpu_math.py.
                            float nd[U]; double z[U], kd[U], r[U], sc[U]; uint64_t ki[U];
                            const double N = 32.0, InvLn2N = 0x1.71547652b82fep+0 * N, Shift = 0x1.8p+52;
                            const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
#pragma unroll
                            for (int u = 0; u < U; u++) {
                                const float fy = (float)(dy0 + jj + u) - ei.yr;
                                const float yy = fy * fy;
                                float dd = xx + yy;
                                dd = dd * P.inv_two_sig2;
                                nd[u] = -dd;
                            }
#pragma unroll
                            for (int u = 0; u < U; u++) { z[u] = InvLn2N * (double)nd[u]; kd[u] = z[u] + Shift; }
#pragma unroll
                            for (int u = 0; u < U; u++) {
                                ki[u] = (uint64_t)__double_as_longlong(kd[u]);
                                uint64_t tt = tab[ki[u] & 31];
                                tt += ki[u] << 47;
                                sc[u] = __longlong_as_double((long long)tt);
                            }
#pragma unroll
                            for (int u = 0; u < U; u++) { kd[u] = kd[u] - Shift; r[u] = z[u] - kd[u]; }
#pragma unroll
                            for (int u = 0; u < U; u++) {
                                const double zz = C0 * r[u] + C1;
                                const double r2 = r[u] * r[u];
                                double y = C2 * r[u] + 1.0;
                                y = zz * r2 + y;
                                y = y * sc[u];
                                const float ev = (float)y;
                                const float q0 = ev * P.rcp_norm;
                                const float r0 = fmaf(-P.norm, q0, ev);
                                v[u] = fmaf(r0, P.rcp_norm, q0);
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < U; u++) {
                                if (MODE == 2) v[u] = 0.001f;
                                else {
                                    const float fy = (float)(dy0 + jj + u) - ei.yr;
                                    const float yy = fy * fy;
                                    float dd = xx + yy;
                                    dd = dd / P.two_sig2;
                                    v[u] = dev_expf_nonpos<true>(-dd, tab) / P.norm;
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            float* dst = on[u] ? vcol + (jj + u) * 8 * kValStride : sink;
                            *dst = POL ? ei.sg * v[u] : v[u];
                        }
                    }
                }
            }
        }
#ifdef EORB_DIAG
        d_work += __builtin_readcyclecounter() - d_s;
#endif
        EORB_TR(6);
        __syncthreads();
    }
#ifdef EORB_DIAG
    if (nbatch > 1000 && lane == 0) {
        const unsigned long long tot = __builtin_readcyclecounter() - d_t0;
        const int role = wave == 0 ? 0 : (wave == 1 ? 1 : 2);
        atomicAdd(&g_diag[role * 4 + 0], d_work);
        atomicAdd(&g_diag[role * 4 + 1], tot);
        atomicAdd(&g_diag[role * 4 + 2], (unsigned long long)nbatch);
        atomicAdd(&g_diag[role * 4 + 3], d_setup);
    }
#endif
    if (wave != 0) return;
    // Without polarity every increment is >= 0: the running maximum is the largest final value, and an add of a tap that
    // underflowed to 0 still counts as a visit (newVal > maxVal, :255), so a tile with any visited pixel offers its pixels' values
    // (0 where nothing was added).  With polarity the extremes were tracked on the real adds above; taps that underflow to 0
    // (sigma < 0.2) would add visits the lists cannot tell from "not touched": the host rejects that configuration.
    if (!POL && any_ok) vmax = fmaxf(vmax, acc);
    if (inimg) img[(size_t)slice * P.W * P.H + (size_t)py * P.W + px] = acc;
    else { vmax = -1000000.0f; vmin = 0.0f; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, d, 64));
        vmin = fminf(vmin, __shfl_xor(vmin, d, 64));
    }
    if (lane == 0) {
        atomicMin(&minmax_enc[slice * 2 + 0], enc_f32(vmin));
        atomicMax(&minmax_enc[slice * 2 + 1], enc_f32(vmax));
    }
}

// K2r: the gather for raw sensor events with a Gaussian stamp (the live configuration).  Lists as in K2 (vals[pixel][e], slot e =
// the entry's position in its 64-entry batch), but a two-stage pipeline without a set-up wave, without masks or sinks, and with no
// LDS hand-off besides the lists:
//   waves 1..NW, values(t)  lane = entry e of the batch, wave w owns NC = 8/NW tile columns (w-1, w-1+NW, ...).  Every lane
//                       decodes ITS OWN entry (8 coalesced bytes; the value waves read the same 512 bytes) and derives the
//                       tile-local tap rectangle.  For each of its columns it writes slot e of ALL 8 pixels of the column: the
//                       stamp value where the pixel is inside the rectangle, +0.0f elsewhere (a column that misses the
//                       rectangle, or a lane without an entry, reads the zeros in front of the table).  So a batch rewrites every
//                       slot of its buffer: nothing is ever cleared, and a store instruction writes 64 consecutive words (no bank
//                       conflicts).  The 8 values of a column are one 32-byte read of the sensor pixel's table row, started at the
//                       stamp row that falls on tile row 0 (the table has 8 floats of slack on either side; rows outside the
//                       rectangle are masked to zero bit-wise).  The lane -> column map is static, so the table reads of batch
//                       t+2 are issued at the end of iteration t and have two barriers to land, and the entry of batch t+3 is
//                       requested just before them: no wave waits for memory.
//   wave 0, adds(t-1)   lane = pixel: acc += list[e], e = 0..63 in order; x + 0.0f == x bit for bit, so the real adds happen in
//                       event order (newVal = image + polSign*val, :251-254).  Reads run 16 slots ahead of the adds.
// NC = 2 (four value waves) has the shortest chain per batch and serves launches that cannot fill the chip; NC = 4 (two value
// waves: the rectangle arithmetic once per four columns) has the fewest instructions per batch and serves the large ones.
constexpr int kStampPad = 8;        // floats of slack in front of (and behind) the stamp table
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
template <bool POL, int NC>
__global__ __launch_bounds__(64 * (1 + 8 / NC)) void ev_gather_raw_kernel(const int64_t* __restrict__ slice_ebase,
                                                                     const int32_t* __restrict__ order, GatherParams P,
                                                                     const uint32_t* __restrict__ tile_cnt,
                                                                     const uint32_t* __restrict__ tile_base,
                                                                     const uint2* __restrict__ entries,
                                                                     float* __restrict__ img, uint32_t* __restrict__ minmax_enc)
{
    // [pixel][entry slot], 64 floats per pixel and no padding: exactly 32 KB, five workgroups per CU.  The 16-byte groups of a
    // pixel's list are stored XOR-swizzled (group g of pixel p at g ^ (p & 15)), so that the add wave's ds_read_b128 (lane =
    // pixel, same logical group) spread over all banks; a value wave's store (lane = slot, one pixel) stays 64 consecutive words.
    __shared__ __attribute__((aligned(16))) float vals[2][64 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // One tile per workgroup: the hardware balances the heaviest-first order dynamically.  (Launches of nearly empty tiles take K2s;
    // a loop over several tiles around this body cost 14 VGPRs and with them the fifth workgroup per CU.)
    bool any_ok = false;                             // (wave 1) some entry of the tile touches an in-image pixel
    const int logical = order[blockIdx.x];
    const int slice = logical / P.NT;
    const int tile = logical - slice * P.NT;
    const int tx0 = (tile % P.TX) * kTile, ty0 = (tile / P.TX) * kTile;
    const int xhi = min(tx0 + kTile - 1, P.W - 1), yhi = min(ty0 + kTile - 1, P.H - 1);
    const int h = P.h, SWP = P.stamp_colstride;
    const int nent = (int)tile_cnt[logical];
    const int nbatch = (nent + 63) >> 6;
    if (nent == 0) {
        // nothing touches the tile: zeros, and no offer to the running extremes (max stays -1e6, min 0: resolveMinMaxVals :32-39)
        if (wave == 0) {
            const int px = tx0 + (lane & 7), py = ty0 + (lane >> 3);
            if (px < P.W && py < P.H) img[(size_t)slice * P.W * P.H + (size_t)py * P.W + px] = 0.0f;
        }
        return;
    }
    const uint2* list = entries + (size_t)slice_ebase[slice] + tile_base[logical];
    constexpr int kRowFloats = 8 * 64;
    float acc = 0.0f, vmax = -1000000.0f, vmin = 0.0f;
    uint32_t x16 = (uint32_t)((lane & 15) << 4);
    // ---- value-wave state: NC tile columns (wave w: columns w-1, w-1+NW, ...); the columns of batches t and t+1 in two register
    //      sets (landed / being loaded), the entry of batch t+2 in registers.  The global loads of this loop are issued from inline
    //      asm with hand-placed s_waitcnt vmcnt(N): vmcnt retires in order, and every iteration issues the same 1 + 2*NC loads in
    //      the same order (entry of batch t+3, then the column reads of batch t+2; dummy addresses past the end of the list), so
    //      "the set of batch t has landed" is vmcnt(2*NC+1) and "the entry of batch t+2 has landed" is vmcnt(2*NC).  (Left to the
    //      compiler, the waits across the loop's back edge degrade to vmcnt(0): a full memory round trip per batch.) ----
    constexpr int NW = 8 / NC;                       // value waves
    const float* const tab0 = P.stamps - kStampPad;  // start of the table's front padding
    // E: entry registers (two sets as well: the load of batch t+3 must not land in registers prepare() is still reading)
    struct ColSet { v4f c[NC][2]; uint32_t m; float sg; v2u E; bool Ev; };        // tile rows 0..7 of the columns, the row mask
    ColSet S0, S1;
#pragma unroll
    for (int k = 0; k < NC; k++) S0.c[k][0] = S0.c[k][1] = S1.c[k][0] = S1.c[k][1] = (v4f){0.f, 0.f, 0.f, 0.f};
    S0.m = S1.m = 0u; S0.sg = S1.sg = 1.0f; S0.E = S1.E = (v2u){0u, 0u}; S0.Ev = S1.Ev = false;
    auto load_entry = [&](int t, ColSet& S) {
        const int j = t * 64 + lane; S.Ev = j < nent;
        const uint2* p = list + min(j, max(nent - 1, 0));
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(S.E) : "v"(p) : "memory");
    };
    // rectangle of an entry, row masks of the columns, then the table reads
    auto prepare = [&](ColSet& S, const v2u ent, const bool valid) {
        const uint32_t w0 = ent.x, w1 = ent.y;
        const int xi = valid ? (int)(int16_t)(w1 & 0xffff) : -32768, yi = (int)(int16_t)(w1 >> 16);
        const int a0 = max(xi - h, tx0) - tx0, a1 = min(xi + h, xhi) - tx0;
        const int b0 = max(yi - h, ty0) - ty0, b1 = min(yi + h, yhi) - ty0;
        const bool ok = a1 >= a0 && b1 >= b0;
        any_ok = any_ok || ok;
        S.m = ok ? ((2u << (b1 & 31)) - 1u) & ~((1u << (b0 & 31)) - 1u) : 0u;     // tile rows b0..b1
        S.sg = (w0 >> 31) ? -1.0f : 1.0f;
        // float index of the table value that falls on tile row 0 of column c: row (ty0 - yi + h) of stamp column (tx0 + c - xi + h)
        const int base = (int)__umul24(w0 & 0x7fffffffu, (uint32_t)P.stamp_stride) + (ty0 - yi + h);
        const bool halfA = b0 <= 3, halfB = b1 >= 4;
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const int cc = wave - 1 + k * NW;
            const bool in = ok && cc >= a0 && cc <= a1;
            // byte offset from the start of the front padding (scalar base + 32-bit lane offset: no 64-bit address arithmetic);
            // a column outside the rectangle reads the 8 zeros there
            const uint32_t off = in ? (uint32_t)(base + kStampPad + (int)__umul24((uint32_t)(tx0 + cc - xi + h), (uint32_t)SWP)) << 2 : 0u;
            // a half of the column (tile rows 0-3 / 4-7) that lies outside the rectangle reads the zeros too: 57 % of the entries
            // touch one half of their tile only, and lanes that share an address share the request
            const uint32_t offA = halfA ? off : 0u, offB = halfB ? off : 0u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(S.c[k][0]) : "v"(offA), "s"(tab0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(S.c[k][1]) : "v"(offB), "s"(tab0) : "memory");
        }
    };
    // wait until all but the n youngest loads have landed; the "+v" operands keep the uses of the set / entry behind the wait
    auto wait_set = [&](ColSet& S) {
        if constexpr (NC == 2)
            asm volatile("s_waitcnt vmcnt(5)" : "+v"(S.c[0][0]), "+v"(S.c[0][1]), "+v"(S.c[1][0]), "+v"(S.c[1][1]) :: "memory");
        else
            asm volatile("s_waitcnt vmcnt(9)" : "+v"(S.c[0][0]), "+v"(S.c[0][1]), "+v"(S.c[1][0]), "+v"(S.c[1][1]),
                         "+v"(S.c[2][0]), "+v"(S.c[2][1]), "+v"(S.c[3][0]), "+v"(S.c[3][1]) :: "memory");
    };
    auto wait_entry = [&](ColSet& S) {
        if constexpr (NC == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(S.E) :: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" : "+v"(S.E) :: "memory");
    };
    if (wave >= 1) {
        // in-flight order expected by the loop: columns(0), entry(2) [in S0.E], columns(1)
        load_entry(0, S0);
        load_entry(1, S1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(S0.E), "+v"(S1.E) :: "memory");
        prepare(S0, S0.E, S0.Ev);
        asm volatile("s_nop 0" : "+v"(S0.E) :: "memory");       // prepare's reads of S0.E are issued before the load below
        load_entry(2, S0);
        prepare(S1, S1.E, S1.Ev);
    }
    __syncthreads();
#ifdef EORB_DIAG
    unsigned long long d_work = 0, d_t0 = __builtin_readcyclecounter(), d_setup = 0;
#endif
    // one iteration: adds(t-1) by wave 0, values(t) by the others from register set S, whose registers then take batch t+2
    auto iteration = [&](int t, ColSet& S, ColSet& T) {      // T: the other set (its entry registers are free)
#ifdef EORB_DIAG
        const unsigned long long d_s = __builtin_readcyclecounter();
#endif
        if (wave == 0) {
            // ---- adds(t-1): all 64 slots in order ----
            if (t >= 1) {
                const char* vb = (const char*)(vals[(t - 1) & 1] + lane * 64);
                // logical group g of this pixel's list: byte offset (g ^ (pixel & 15)) * 16.  x16 is made opaque per iteration:
                // hoisted out of the loop the 16 offsets would cost 16 VGPRs (and with them the fifth workgroup per CU)
                asm volatile("" : "+v"(x16));
#define EORB_RD(g) (*(const float4*)(vb + (((uint32_t)(g) << 4) ^ x16)))
                float4 q[4] = {EORB_RD(0), EORB_RD(1), EORB_RD(2), EORB_RD(3)};
#pragma unroll
                for (int g = 0; g < 16; g += 4) {
                    float4 n[4];
                    if (g + 4 < 16) { n[0] = EORB_RD(g + 4); n[1] = EORB_RD(g + 5); n[2] = EORB_RD(g + 6); n[3] = EORB_RD(g + 7); }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (!POL) { acc = acc + q[u].x; acc = acc + q[u].y; acc = acc + q[u].z; acc = acc + q[u].w; }
                        else {
                            // running extremes (resolveMinMaxVals :32-39) only move on a real add: a tap is never 0
#define EORB_ADD(val) { if ((val) != 0.0f) { acc = acc + (val); vmax = fmaxf(vmax, acc); vmin = fminf(vmin, acc); } }
                            EORB_ADD(q[u].x) EORB_ADD(q[u].y) EORB_ADD(q[u].z) EORB_ADD(q[u].w)
#undef EORB_ADD
                        }
                    }
                    if (g + 4 < 16) { q[0] = n[0]; q[1] = n[1]; q[2] = n[2]; q[3] = n[3]; }
                }
#undef EORB_RD
            }
        } else if (t < nbatch) {
            // ---- values(t): the columns requested two iterations ago; slot `lane` of the 8 pixels of each column ----
            wait_set(S);
            // value where bit r of the entry's row mask is set, +0.0f elsewhere: a sign-extended bit (asm: v_bfe_i32) as an AND mask,
            // shared by the wave's columns
            uint32_t rm[8];
#define EORB_RM(r) asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(rm[r]) : "v"(S.m), "i"(r));
            EORB_RM(0) EORB_RM(1) EORB_RM(2) EORB_RM(3) EORB_RM(4) EORB_RM(5) EORB_RM(6) EORB_RM(7)
#undef EORB_RM
            // pixel (c, r) = c + 8r: its swizzle (p & 15) = c | (r & 1) << 3, i.e. odd rows flip bit 3 of the group index (32 floats)
#define EORB_ROW(dst, r, val) ((r) & 1 ? (dst) + 32 * ((lane & 32) ? -1 : 1) : (dst))[(r) * kRowFloats] = __uint_as_float(__float_as_uint(POL ? S.sg * (val) : (val)) & rm[r]);
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int cc = wave - 1 + k * NW;
                float* d = vals[t & 1] + cc * 64 + ((((lane >> 2) ^ cc) << 2) | (lane & 3));
                EORB_ROW(d, 0, S.c[k][0].x) EORB_ROW(d, 1, S.c[k][0].y) EORB_ROW(d, 2, S.c[k][0].z) EORB_ROW(d, 3, S.c[k][0].w)
                EORB_ROW(d, 4, S.c[k][1].x) EORB_ROW(d, 5, S.c[k][1].y) EORB_ROW(d, 6, S.c[k][1].z) EORB_ROW(d, 7, S.c[k][1].w)
            }
#undef EORB_ROW
#ifdef EORB_DIAG
            d_setup += __builtin_readcyclecounter() - d_s;
#endif
            {
                // entry of batch t+2 (requested an iteration ago, four loads younger than it in flight); then request the entry of
                // batch t+3 BEFORE the column reads of batch t+2, so that the next iteration's wait for it leaves those in flight
                wait_entry(S);
                load_entry(t + 3, T);
                prepare(S, S.E, S.Ev);
            }
        }
#ifdef EORB_DIAG
        d_work += __builtin_readcyclecounter() - d_s;
#endif
        __syncthreads();
    };
    for (int t = 0; t < nbatch + 1; t += 2) {
        iteration(t, S0, S1);
        if (t + 1 < nbatch + 1) iteration(t + 1, S1, S0);
    }
    // The loads issued from the inline asm run ahead of the batches (entry of batch t + 3, columns of batch t + 2): the last ones are
    // still in flight here, and the compiler, which takes an asm's outputs as written when the asm ends, is free to reuse their target
    // registers from now on.  A late return would then overwrite whatever lives there (the flag below, the next tile's addresses).
    // Drain them while the registers are still theirs.
    if (wave >= 1) {
        if constexpr (NC == 2)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(S0.c[0][0]), "+v"(S0.c[0][1]), "+v"(S0.c[1][0]), "+v"(S0.c[1][1]), "+v"(S0.E),
                         "+v"(S1.c[0][0]), "+v"(S1.c[0][1]), "+v"(S1.c[1][0]), "+v"(S1.c[1][1]), "+v"(S1.E) :: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(S0.c[0][0]), "+v"(S0.c[0][1]), "+v"(S0.c[1][0]), "+v"(S0.c[1][1]),
                         "+v"(S0.c[2][0]), "+v"(S0.c[2][1]), "+v"(S0.c[3][0]), "+v"(S0.c[3][1]), "+v"(S0.E),
                         "+v"(S1.c[0][0]), "+v"(S1.c[0][1]), "+v"(S1.c[1][0]), "+v"(S1.c[1][1]),
                         "+v"(S1.c[2][0]), "+v"(S1.c[2][1]), "+v"(S1.c[3][0]), "+v"(S1.c[3][1]), "+v"(S1.E) :: "memory");
    }
#ifdef EORB_DIAG
    if (nbatch > 1000 && lane == 0) {
        const unsigned long long tot = __builtin_readcyclecounter() - d_t0;
        const int role = wave == 0 ? 0 : (wave == 1 ? 1 : 2);
        atomicAdd(&g_diag[role * 4 + 0], d_work);
        atomicAdd(&g_diag[role * 4 + 1], tot);
        atomicAdd(&g_diag[role * 4 + 2], (unsigned long long)nbatch);
        atomicAdd(&g_diag[role * 4 + 3], d_setup);
    }
#endif
    // the flag of wave 1 travels through the (now free) list buffer
    if (wave == 1) { const bool f = __any(any_ok); if (lane == 0) vals[0][0] = f ? 1.0f : 0.0f; }
    __syncthreads();
    if (wave == 0) {
    const bool tile_ok = vals[0][0] != 0.0f;
    const int lx = lane & 7, ly = lane >> 3;
    const int px = tx0 + lx, py = ty0 + ly;
    const bool inimg = px < P.W && py < P.H;
    // Without polarity every increment is >= 0: the running maximum is the largest final value, and an add of a tap that
    // underflowed to 0 still counts as a visit (newVal > maxVal, :255), so a tile with any visited pixel offers its pixels' values
    // (0 where nothing was added).  With polarity the extremes were tracked on the real adds above; taps that underflow to 0
    // (sigma < 0.2) would add visits the lists cannot tell from "not touched": the host rejects that configuration.
    if (!POL && tile_ok) vmax = fmaxf(vmax, acc);
    if (inimg) img[(size_t)slice * P.W * P.H + (size_t)py * P.W + px] = acc;
    else { vmax = -1000000.0f; vmin = 0.0f; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, d, 64));
        vmin = fminf(vmin, __shfl_xor(vmin, d, 64));
    }
    if (lane == 0) {
        atomicMin(&minmax_enc[slice * 2 + 0], enc_f32(vmin));
        atomicMax(&minmax_enc[slice * 2 + 1], enc_f32(vmax));
    }
    }
}

// K2s: the same gather for launches whose tiles are nearly empty (the 2 000-event slices of the live event path leave a tile about
// nine entries).  K2r's pipeline costs a tile several memory round trips and two barriers whatever its list holds; here ONE wave
// owns a tile, without LDS and without barriers (32 waves per CU hide each other's latency): lane = pixel, the tile's entries are
// read 64 at a time (lane = entry) and handed round by v_readlane, and every lane reads its own tap of the entry's stamp -- the
// 8 x 8 window of a stamp is 8 runs of 8 consecutive floats -- and adds it if the stamp reaches its pixel.  Adds are in list order;
// a lane the stamp does not reach keeps its value, which is what K2r's + 0.0f does.
constexpr int kSparseWaves = 4;
template <bool POL>
__global__ __launch_bounds__(64 * kSparseWaves) void ev_gather_sparse_kernel(const int64_t* __restrict__ slice_ebase, GatherParams P, int per_wave,
                                                                            const uint32_t* __restrict__ tile_cnt, const uint32_t* __restrict__ tile_base,
                                                                            const uint2* __restrict__ entries, float* __restrict__ img,
                                                                            uint32_t* __restrict__ minmax_enc)
{
    const int lane = threadIdx.x & 63;
    const int first = (blockIdx.x * kSparseWaves + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * per_wave;
    const int SW = 2 * P.h + 1, SWP = P.stamp_colstride;
    // A work item is a chain of dependent loads (count -> entries -> stamp taps) for a handful of adds.  The wave reads the counts and
    // list places of ALL its items at once (lane = item, per_wave <= 64) and the first 64 entries of item k + 1 while it works on item k.
    const int n_here = min(per_wave, P.total - first);
    uint32_t cnt_v = 0u, off_lo = 0u, off_hi = 0u;
    if (lane < n_here) {
        const int li = first + lane;
        cnt_v = tile_cnt[li];
        const uint64_t o = (uint64_t)slice_ebase[li / P.NT] + tile_base[li];
        off_lo = (uint32_t)o; off_hi = (uint32_t)(o >> 32);
    }
    auto list_of = [&](int k) {
        return entries + (((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)off_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)off_lo, k));
    };
    uint2 ahead = make_uint2(0u, 0u);
    if (n_here > 0) { const int n0 = __builtin_amdgcn_readlane((int)cnt_v, 0); if (n0) ahead = list_of(0)[min(lane, min(64, n0) - 1)]; }
    // (slice, tile column, tile row) of the items by increments: three integer divisions per item were a quarter of its instructions
    int slice = first / P.NT, tx = (first - slice * P.NT) % P.TX, ty = (first - slice * P.NT) / P.TX;
    tx -= 1;
    for (int k = 0; k < n_here; k++) {
        if (++tx == P.TX) { tx = 0; if (++ty == P.NT / P.TX) { ty = 0; slice++; } }
        const int tx0 = tx * kTile, ty0 = ty * kTile;
        const int px = tx0 + (lane & 7), py = ty0 + (lane >> 3);
        const bool inimg = px < P.W && py < P.H;
        float* const dst = img + (size_t)slice * P.W * P.H + (size_t)py * P.W + px;
        const int nent = __builtin_amdgcn_readlane((int)cnt_v, k);
        const uint2 first64 = ahead;
        if (k + 1 < n_here) { const int n1 = __builtin_amdgcn_readlane((int)cnt_v, k + 1); if (n1) ahead = list_of(k + 1)[min(lane, min(64, n1) - 1)]; }
        if (nent == 0) { if (inimg) *dst = 0.0f; continue; }      // no offer to the running extremes (resolveMinMaxVals :32-39)
        const uint2* list = list_of(k);
        float acc = 0.0f, vmax = -1000000.0f, vmin = 0.0f;
        bool touched = false;                                   // this (in-image) pixel was reached by some stamp
        for (int e0 = 0; e0 < nent; e0 += 64) {
            const int cnt = min(64, nent - e0);
            const uint2 mine = e0 == 0 ? first64 : list[e0 + min(lane, cnt - 1)];
            constexpr int U = 8;
            for (int k0 = 0; k0 < cnt; k0 += U) {
                float v[U]; bool in[U]; uint32_t sgn[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int k = min(k0 + u, cnt - 1);
                    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, k), w1 = (uint32_t)__builtin_amdgcn_readlane((int)mine.y, k);
                    const int xi = (int)(int16_t)(w1 & 0xffff), yi = (int)(int16_t)(w1 >> 16);
                    const uint32_t i = (uint32_t)(px - xi + P.h), j = (uint32_t)(py - yi + P.h);
                    in[u] = (k0 + u < cnt) && i < (uint32_t)SW && j < (uint32_t)SW && inimg;
                    sgn[u] = w0 & 0x80000000u;
                    const uint32_t off = (w0 & 0x7fffffffu) * (uint32_t)P.stamp_stride + i * (uint32_t)SWP + j;
                    v[u] = in[u] ? P.stamps[off] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (in[u]) {
                        acc = acc + (POL ? __uint_as_float(__float_as_uint(v[u]) ^ sgn[u]) : v[u]);
                        if (POL) { vmax = fmaxf(vmax, acc); vmin = fminf(vmin, acc); }
                        touched = true;
                    }
            }
        }
        // as in K2r: without polarity a tile some stamp reaches inside the image offers its pixels' values (0 where nothing was added)
        const bool tile_ok = __any(touched);
        if (!POL && tile_ok) vmax = fmaxf(vmax, acc);
        if (inimg) *dst = acc;
        else { vmax = -1000000.0f; vmin = 0.0f; }
        // (DPP reductions, result in lane 63; without polarity the minimum never leaves its initial 0: ev_minmax_init_kernel)
        vmax = wave_max_to_lane63(vmax);
        if (POL) vmin = wave_min_to_lane63(vmin);
        if (lane == 63) {
            if (POL) atomicMin(&minmax_enc[slice * 2 + 0], enc_f32(vmin));
            atomicMax(&minmax_enc[slice * 2 + 1], enc_f32(vmax));
        }
    }
}

// K2d: one or a few SMALL slices per call (the live path: eorb_ev2im_gauss_raw on a 2 000-event slice).  Binning costs four launches
// and a descriptor upload there; instead every tile's wave reads ALL events of its slice (690 waves x 2 000 events: nothing),
// keeps those whose stamp reaches the tile -- in event order, ballot-compacted into a small LDS list -- and adds them as K2s does.
// Same entries in the same order as the binned lists, hence the same image.
constexpr int kDirectSlices = 8;
struct DirectSlices { int64_t beg[kDirectSlices], end[kDirectSlices]; int64_t tab_base; };      // the slices' event ranges (they may overlap: the contest's later-half histogram)
constexpr int kDirectList = 2048;                   // entries of the LDS list (flushed when the next batch of loads might not fit)
#ifndef EORB_DIRECT_G
#define EORB_DIRECT_G 8
#endif
// ev_direct_slices_dev resolves every event of the call ONCE, in parallel, into the entry it would contribute to a tile's list
// (ev_pre_kernel): { table row | negative polarity << 31, xi | yi << 16 } -- raw sensor events: row = the sensor pixel (its stamp in
// the maps' table), integer position from src_info; float events (eorb_event16: the reference's own seam, ev2im_gauss(vector<
// EventData>)): row = the event itself, whose integer position and (2h+1)^2 taps are tabulated per EVENT by the same kernels that
// tabulate them per sensor pixel (ev_stamp_kernel: the arithmetic of exp_XY2f :59-65 as every other path evaluates it).  The tile
// wavefronts then only test and copy: with the map lookup (raw) or an f64 expf chain per listed entry (float) inside this kernel a
// 2 000 ... 5 000-event chunk took 76 / 140 us.
template <bool FLT>
__global__ void ev_pre_kernel(const eorb_raw_event* __restrict__ ev, int n, int W, int H, int LW, int LH, const uint32_t* __restrict__ src_info,
                              uint32_t row0, uint2* __restrict__ pre, uint32_t* __restrict__ mm_init = nullptr, int nmm = 0)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    // (the slices' running extremes start here when the caller did not send them initialised: no launch of their own)
    if (mm_init && k < nmm) { mm_init[2 * k] = enc_f32(0.0f); mm_init[2 * k + 1] = enc_f32(-1000000.0f); }
    if (k >= n) return;
    uint32_t row, neg, info = 0x80008000u;
    if (FLT) {
        const uint4 q = *(const uint4*)&ev[k];                             // { x, y } float bits, t
        const float ex = __uint_as_float(q.x), ey = __uint_as_float(q.y);
        row = row0 + (uint32_t)k; neg = q.w >> 31;                           // negative polarity = the sign bit of t (eorb_pack_events)
        if (ex == ex && ey == ey) {                                          // NaN coordinates are never in the image (ev_tile_range)
            const int xi = (int)fminf(fmaxf(floorf(ex), -32000.f), 32000.f), yi = (int)fminf(fmaxf(floorf(ey), -32000.f), 32000.f);   // breakFloatCoords :51-57
            info = (uint32_t)(xi & 0xffff) | ((uint32_t)(yi & 0xffff) << 16);
        }
    } else {
        const uint2 q = *(const uint2*)&ev[k];                             // { x | y << 16, p }
        const uint32_t x = q.x & 0xffff, y = q.x >> 16;
        row = y * (uint32_t)LW + x; neg = q.y ? 0u : 1u;
        if (x < (uint32_t)LW && y < (uint32_t)LH) info = src_info[row]; else row = 0u;
    }
    pre[k] = make_uint2(row | (neg << 31), info);
}

template <bool POL>
__global__ __launch_bounds__(64) void ev_gather_direct_kernel(const uint2* __restrict__ pre, DirectSlices S, GatherParams P,
                                                              float* __restrict__ img, uint32_t* __restrict__ minmax_enc)
{
    __shared__ uint2 lst[kDirectList];
    const int lane = threadIdx.x;
    const int slice = blockIdx.x / P.NT, tile = blockIdx.x - slice * P.NT;
    const int tx0 = (tile % P.TX) * kTile, ty0 = (tile / P.TX) * kTile;
    const int px = tx0 + (lane & 7), py = ty0 + (lane >> 3);
    const bool inimg = px < P.W && py < P.H;
    const int SW = 2 * P.h + 1, SWP = P.stamp_colstride;
    const uint2* e = pre + (S.beg[slice] - S.tab_base);
    const int n = (int)(S.end[slice] - S.beg[slice]);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    float acc = 0.0f, vmax = -1000000.0f, vmin = 0.0f;
    bool touched = false, any = false;
    int nl = 0;
    // the adds of the listed entries (K2s' loop over a list in LDS).  The taps of a group of 16 entries are requested while the previous
    // group's are added.
    auto flush = [&]() {
        constexpr int U = 16;
        for (int e0 = 0; e0 < nl; e0 += 64) {
            const int cnt = min(64, nl - e0);
            const uint2 mine = lst[e0 + min(lane, cnt - 1)];
            float v[U]; bool in[U]; uint32_t sgn[U];
            auto fetch = [&](int k0) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int k = min(k0 + u, cnt - 1);
                    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, k), w1 = (uint32_t)__builtin_amdgcn_readlane((int)mine.y, k);
                    const int xi = (int)(int16_t)(w1 & 0xffff), yi = (int)(int16_t)(w1 >> 16);
                    const uint32_t i = (uint32_t)(px - xi + P.h), j = (uint32_t)(py - yi + P.h);
                    in[u] = (k0 + u < cnt) && i < (uint32_t)SW && j < (uint32_t)SW && inimg;
                    sgn[u] = w0 & 0x80000000u;
                    const uint32_t off = (w0 & 0x7fffffffu) * (uint32_t)P.stamp_stride + i * (uint32_t)SWP + j;
                    v[u] = in[u] ? P.stamps[off] : 0.0f;
                }
            };
            fetch(0);
            for (int k0 = 0; k0 < cnt; k0 += U) {
                float a[U]; bool ain[U]; uint32_t asg[U];
#pragma unroll
                for (int u = 0; u < U; u++) { a[u] = v[u]; ain[u] = in[u]; asg[u] = sgn[u]; }
                if (k0 + U < cnt) fetch(k0 + U);
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (ain[u]) {
                        acc = acc + (POL ? __uint_as_float(__float_as_uint(a[u]) ^ asg[u]) : a[u]);
                        if (POL) { vmax = fmaxf(vmax, acc); vmin = fminf(vmin, acc); }
                        touched = true;
                    }
            }
        }
        nl = 0;
    };
    constexpr int G = EORB_DIRECT_G;                         // sub-batches of 64 events whose loads are in flight together
#ifdef EORB_DIRECT_TIMING
    const long long td0 = clock64(); long long td_load = 0, td_flush = 0;
#endif
    for (int k0 = 0; k0 < n; k0 += 64 * G) {
#ifdef EORB_DIRECT_TIMING
        const long long tl0 = clock64();
#endif
        uint2 q[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int k = k0 + g * 64 + lane;
            q[g] = k < n ? e[k] : make_uint2(0u, 0x80008000u);
        }
#ifdef EORB_DIRECT_TIMING
        if (q[G - 1].y == 0x12345678u) any = true;            // (wait for the loads)
        td_load += clock64() - tl0;
#endif
        if (nl + 64 * G > kDirectList) flush();                 // (one call site: the adds' code is long, and a copy per sub-batch was paid in instruction fetches)
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (k0 + g * 64 >= n) break;
            const int xi = (int)(int16_t)(q[g].y & 0xffff), yi = (int)(int16_t)(q[g].y >> 16);
            // the event has an entry in this tile's list iff the tile lies in its tile range (ev_tile_range / ev_tile_range_raw):
            // tx0 - h <= xi <= tx0 + kTile - 1 + h, the same in y -- one unsigned compare per axis; a dropped event, (-32768, -32768), is far
            // outside every tile (this wavefront is alone on its SIMD most of the time: the scan costs its instruction count)
            const bool hit = (uint32_t)(xi - (tx0 - P.h)) < (uint32_t)(kTile + 2 * P.h) && (uint32_t)(yi - (ty0 - P.h)) < (uint32_t)(kTile + 2 * P.h);
            const uint64_t m = __ballot(hit);
            if (m) {
                if (hit) lst[nl + __popcll(m & lt_mask)] = q[g];
                nl += __popcll(m); any = true;
            }
        }
    }
#ifdef EORB_DIRECT_TIMING
    const long long tf0 = clock64(); const int nl_last = nl;
#endif
    flush();
#ifdef EORB_DIRECT_TIMING
    td_flush = clock64() - tf0;
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 345 || blockIdx.x == 689)) printf("direct gather wg %d: n %d, listed %d: loads %lld | scan %lld | adds %lld (ticks at 2.4 per ns)\n", (int)blockIdx.x, n, nl_last, td_load, tf0 - td0 - td_load, td_flush);
#endif
    float* const dst = img + (size_t)slice * P.W * P.H + (size_t)py * P.W + px;
    if (!any) { if (inimg) *dst = 0.0f; return; }          // an empty list: no offer to the running extremes (as K2s / K2r)
    const bool tile_ok = __any(touched);
    if (!POL && tile_ok) vmax = fmaxf(vmax, acc);
    if (inimg) *dst = acc;
    else { vmax = -1000000.0f; vmin = 0.0f; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, d, 64));
        vmin = fminf(vmin, __shfl_xor(vmin, d, 64));
    }
    if (lane == 0) {
        atomicMin(&minmax_enc[slice * 2 + 0], enc_f32(vmin));
        atomicMax(&minmax_enc[slice * 2 + 1], enc_f32(vmax));
    }
}

// exhaustive self-check helper: IEEE quotient vs the reciprocal/fma sequence used above
__global__ void ev_divcheck_kernel(uint32_t lo_bits, uint32_t hi_bits, float norm, float rcp, unsigned long long* bad)
{
    unsigned long long local = 0;
    for (uint64_t u = (uint64_t)lo_bits + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; u <= hi_bits;
         u += (uint64_t)gridDim.x * blockDim.x) {
        const float ev = __uint_as_float((uint32_t)u);
        const float q0 = ev * rcp;
        const float r0 = fmaf(-norm, q0, ev);
        const float q = fmaf(r0, rcp, q0);
        const float ref = ev / norm;
        if (__float_as_uint(q) != __float_as_uint(ref)) local++;
    }
    if (local) atomicAdd(bad, local);
}

__global__ void ev_minmax_init_kernel(uint32_t* mm, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { mm[2 * i] = enc_f32(0.0f); mm[2 * i + 1] = enc_f32(-1000000.0f); }
}

// normalizeImage :67-72 -> Mat::convertTo(CV_8UC1, alpha, beta) (SURVEY App.B H2)
// only_if_range: ev2im normalises only when max > min (:206)
__global__ void ev_normalize_kernel(const float* __restrict__ img, const uint32_t* __restrict__ mm,
                                    uint8_t* __restrict__ out, int npix, int only_if_range)
{
    const int slice = blockIdx.y;
    const float mn = dec_f32(mm[2 * slice]), mx = dec_f32(mm[2 * slice + 1]);
    if (only_if_range && !(mx > mn)) return;
    const float alpha = 255.f / (mx - mn);
    const float beta = -mn * alpha;
    const float* src = img + (size_t)slice * npix;
    uint8_t* dst = out + (size_t)slice * npix;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const float m = src[i] * alpha;
        const float v = m + beta;
        int iv = __float2int_rn(v);
        iv = min(max(iv, 0), 255);
        dst[i] = (uint8_t)iv;
    }
}

__global__ void ev_decode_minmax_kernel(const uint32_t* mm, float* out, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * B) out[i] = dec_f32(mm[i]);
}

// ---------------------------------------------------------------------------------------------------
// Motion-compensated accumulation: ev2mci_gg_f (src/Event/EventConversion.cc:280-531) = a per-event warp
// followed by exactly the ev2im_gauss splat.  The warp kernels rewrite (x, y) of the 16-byte records; the splat is the
// pipeline above.
// GeometricCamera of the warp: model 0 = Pinhole (CameraModels/Pinhole.cpp:30-62), 1 = KannalaBrandt8 (KannalaBrandt8.cpp:87-190)
struct WarpCam { int model; float fx, fy, cx, cy, k0, k1, k2, k3, precision; };

// pCamera->unproject(cv::Point2f) -> (X, Y, 1)
__device__ __forceinline__ void cam_unproject(const WarpCam& c, float x, float y, float& X, float& Y)
{
    const float pwx = (x - c.cx) / c.fx, pwy = (y - c.cy) / c.fy;
    if (c.model == 0) { X = pwx; Y = pwy; return; }
    // Newton iterations on theta, all in float (:164-187)
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf((float)(-3.1415926535897932384626433832795 / 2.f), theta_d), (float)(3.1415926535897932384626433832795 / 2.f));
    if ((double)theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = c.k0 * theta2, k1_theta4 = c.k1 * theta4;
            const float k2_theta6 = c.k2 * theta6, k3_theta8 = c.k3 * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < c.precision) break;
        }
        scale = dev_tanf(theta) / theta_d;
    }
    X = pwx * scale; Y = pwy * scale;
}

struct WarpSE3 {
    WarpCam cam;
    double angle, ax, ay, az, tx, ty, tz;
    float medDepth;
};

__global__ void ev_warp_se3_kernel(const eorb_event16* __restrict__ in, eorb_event16* __restrict__ out, int n, WarpSE3 P,
                                   const float* __restrict__ depth)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double t1 = fabs(in[n - 1].t), t0 = fabs(in[0].t);
    const double DT = t1 - t0;
    const double invDT = 1.0 / DT;
    const eorb_event16 e = in[k];
    const double etRate = (t1 - fabs(e.t)) * invDT;
    float X, Y;
    cam_unproject(P.cam, e.x, e.y, X, Y);
    const double Pv[3] = {(double)X, (double)Y, 1.0};
    const double a = P.angle * etRate;
    double sn, c;
    dev_dsincos(a, &sn, &c);
    // Eigen::AngleAxisd::toRotationMatrix()
    const double sax = sn * P.ax, say = sn * P.ay, saz = sn * P.az;
    const double c1x = (1.0 - c) * P.ax, c1y = (1.0 - c) * P.ay, c1z = (1.0 - c) * P.az;
    double R[3][3];
    double tmp;
    tmp = c1x * P.ay; R[0][1] = tmp - saz; R[1][0] = tmp + saz;
    tmp = c1x * P.az; R[0][2] = tmp + say; R[2][0] = tmp - say;
    tmp = c1y * P.az; R[1][2] = tmp - sax; R[2][1] = tmp + sax;
    R[0][0] = c1x * P.ax + c; R[1][1] = c1y * P.ay + c; R[2][2] = c1z * P.az + c;
    const double d = (double)(depth ? depth[k] : P.medDepth);
    const double tt[3] = {P.tx, P.ty, P.tz};
    double np[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        // Eigen 3.3 fixed-size product: row(i).cwiseProduct(P3D).sum() unrolls as a0 + (a1 + a2)
        const double a0 = (d * R[i][0]) * Pv[0];
        const double a1 = (d * R[i][1]) * Pv[1];
        const double a2 = (d * R[i][2]) * Pv[2];
        const double acc = a0 + (a1 + a2);
        np[i] = acc + tt[i] * etRate;
    }
    double u, v;
    if (P.cam.model == 0) {
        u = (double)P.cam.fx * np[0] / np[2] + (double)P.cam.cx;            // Pinhole::project(Eigen::Vector3d)
        v = (double)P.cam.fy * np[1] / np[2] + (double)P.cam.cy;
    } else {
        // KannalaBrandt8::project(Eigen::Vector3d) (:111-133): atan2f / sqrtf on the float-converted arguments, the rest in double
        const double x2_plus_y2 = np[0] * np[0] + np[1] * np[1];
        const double theta = (double)dev_atan2f(sqrtf((float)x2_plus_y2), (float)np[2]);
        const double psi = (double)dev_atan2f((float)np[1], (float)np[0]);
        const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
        const double r = theta + (double)P.cam.k0 * theta3 + (double)P.cam.k1 * theta5 + (double)P.cam.k2 * theta7 + (double)P.cam.k3 * theta9;
        double ps, pc;
        dev_dsincos(psi, &ps, &pc);
        u = (double)P.cam.fx * r * pc + (double)P.cam.cx;
        v = (double)P.cam.fy * r * ps + (double)P.cam.cy;
    }
    eorb_event16 o = e;
    o.x = (float)u; o.y = (float)v;
    out[k] = o;
}

struct WarpSE2 { WarpCam cam; float p0, p1, p2, sc; };

__global__ void ev_warp_se2_kernel(const eorb_event16* __restrict__ in, eorb_event16* __restrict__ out, int n, WarpSE2 P)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double t1 = fabs(in[n - 1].t), t0 = fabs(in[0].t);
    const float DT = (float)(t1 - t0);
    const float invDT = 1.f / DT;
    const float omega0 = P.p0 * invDT, vx0 = P.p1 * invDT, vy0 = P.p2 * invDT;
    const float scDiff = 1.f - P.sc;
    const eorb_event16 e = in[k];
    const float tk = (float)(t1 - fabs(e.t));
    float X, Y;
    cam_unproject(P.cam, e.x, e.y, X, Y);
    const float theta_k = tk * omega0;
    const float currSc = scDiff * (1 - tk * invDT) + P.sc;
    float sn, cs;
    dev_sincosf(theta_k, &sn, &cs);
    const float xp = currSc * (X * cs - Y * sn) + vx0 * tk;
    const float yp = currSc * (X * sn + Y * cs) + vy0 * tk;
    eorb_event16 o = e;
    if (P.cam.model == 0) {
        o.x = P.cam.fx * xp / 1.f + P.cam.cx;                                // Pinhole::project(cv::Point3f)
        o.y = P.cam.fy * yp / 1.f + P.cam.cy;
    } else {
        // KannalaBrandt8::project(cv::Point3f(xp, yp, 1.f)) (:87-103), float throughout
        const float x2_plus_y2 = xp * xp + yp * yp;
        const float theta = dev_atan2f(sqrtf(x2_plus_y2), 1.f);
        const float psi = dev_atan2f(yp, xp);
        const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
        const float r = theta + P.cam.k0 * theta3 + P.cam.k1 * theta5 + P.cam.k2 * theta7 + P.cam.k3 * theta9;
        float ps, pc;
        dev_sincosf(psi, &ps, &pc);
        o.x = P.cam.fx * r * pc + P.cam.cx;
        o.y = P.cam.fy * r * ps + P.cam.cy;
    }
    out[k] = o;
}

static WarpCam warp_cam(const eorb_camera* cam)
{
    return WarpCam{cam->model, cam->fx, cam->fy, cam->cx, cam->cy, cam->k[0], cam->k[1], cam->k[2], cam->k[3], cam->precision};
}

int ev_warp_se3_dev(eorb_ctx* c, const eorb_event16* d_in, eorb_event16* d_out, int n, const eorb_camera* cam, double angle,
                    const double axis[3], const double tt[3], float medDepth, const float* d_depth)
{
    if (n <= 0) return EORB_OK;
    WarpSE3 P{warp_cam(cam), angle, axis[0], axis[1], axis[2], tt[0], tt[1], tt[2], medDepth};
    ProfScope ps(c, "ev_warp_se3");
    ev_warp_se3_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(d_in, d_out, n, P, d_depth);
    EORB_LAUNCH_CHECK(c, "ev_warp_se3_kernel");
    return EORB_OK;
}

int ev_warp_se2_dev(eorb_ctx* c, const eorb_event16* d_in, eorb_event16* d_out, int n, const eorb_camera* cam, const float* params, int nparams)
{
    if (n <= 0) return EORB_OK;
    WarpSE2 P{warp_cam(cam), params[0], params[1], params[2], nparams > 3 ? params[3] : 1.f};
    ProfScope ps(c, "ev_warp_se2");
    ev_warp_se2_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(d_in, d_out, n, P);
    EORB_LAUNCH_CHECK(c, "ev_warp_se2_kernel");
    return EORB_OK;
}

// EvImConverter::measureImageFocus (:74-111): the standard deviation of every 30x30 patch with cv::meanStdDev's raster-order double
// accumulation (the order of those 900 additions is part of the result), then the patch deviations summed in patch order.  One
// wavefront per patch stages the patch in LDS (coalesced rows), one lane walks it in raster order; a second small launch adds the
// patches up (one thread per patch reading 900 floats from global memory one after the other took 93 us; this takes ~10).
__global__ __launch_bounds__(64) void ev_focus_patch_kernel(const float* __restrict__ imgs, int W, int H, int np, float* __restrict__ sd)
{
    // one wavefront per (image, patch): the patch staged in LDS, then cv::meanStdDev's two raster-order f64 sums (sum and sum of
    // squares are independent chains: lane 0 walks one, lane 1 the other)
    __shared__ __attribute__((aligned(16))) float px[30 * 30 + 12];
    __shared__ double s_sum[2];
    const int patch = 30;
    const int pc = (W + patch - 1) / patch;
    const int im = blockIdx.x / np, p = blockIdx.x % np;
    const float* img = imgs + (size_t)im * W * H;
    const int i = (p / pc) * patch, j = (p % pc) * patch;
    const int maxRow = min(i + patch, H), maxCol = min(j + patch, W);
    const int pw = maxCol - j, ph = maxRow - i, n = pw * ph;
    for (int k = threadIdx.x; k < n; k += 64) { const int y = k / pw, x = k - y * pw; px[k] = img[(size_t)(i + y) * W + j + x]; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const bool sq = threadIdx.x == 1;
        double s = 0;
        int k = 0;
        for (; k + 8 <= n; k += 8) {                 // (two 16-byte LDS reads in flight per round; the additions stay in raster order)
            const float4 a = *(const float4*)&px[k], b = *(const float4*)&px[k + 4];
            const float v8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int u = 0; u < 8; u++) { const double v = (double)v8[u]; s += sq ? v * v : v; }
        }
        for (; k < n; k++) { const double v = (double)px[k]; s += sq ? v * v : v; }
        s_sum[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double N = (double)ph * (double)pw;
        const double scale = 1.0 / N;
        const double mean = s_sum[0] * scale;
        double var = s_sum[1] * scale - mean * mean;
        if (var < 0) var = 0;
        sd[blockIdx.x] = (float)sqrt(var);
    }
}
__global__ void ev_focus_sum_kernel(const float* __restrict__ sd, int np, float* __restrict__ out)
{
    if (threadIdx.x == 0) {                          // one block per image
        float localStd = 0.f;
        for (int p = 0; p < np; p++) localStd += sd[(size_t)blockIdx.x * np + p];
        out[blockIdx.x] = localStd / (float)np;
    }
}

// nimg images of W x H back to back (the motion-compensation contest scores its reconstructions together, EvImBuilder.cpp:1165-1203)
int ev_focus_dev(eorb_ctx* c, const float* d_img, int nimg, int W, int H, float* d_out)
{
    const int np = ((W + 29) / 30) * ((H + 29) / 30);
    int rc;
    if ((rc = ensure(c, c->focus_sd, sizeof(float) * (size_t)np * nimg))) return rc;
    ProfScope ps(c, "ev_focus");
    ev_focus_patch_kernel<<<np * nimg, 64, 0, c->stream>>>(d_img, W, H, np, (float*)c->focus_sd.p);
    ev_focus_sum_kernel<<<nimg, 64, 0, c->stream>>>((const float*)c->focus_sd.p, np, d_out);
    EORB_LAUNCH_CHECK(c, "ev_focus kernels");
    return EORB_OK;
}

// cv::normalize(img, img, 255, 0, NORM_MINMAX, CV_8UC1) (EvImBuilder.cpp:1076): min/max of the final image
__global__ void ev_minmax_final_kernel(const float* __restrict__ img, int npix, uint32_t* __restrict__ mm)
{
    float lo = img[0], hi = img[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) { lo = fminf(lo, img[i]); hi = fmaxf(hi, img[i]); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo = fminf(lo, __shfl_xor(lo, d, 64)); hi = fmaxf(hi, __shfl_xor(hi, d, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], enc_f32(lo)); atomicMax(&mm[1], enc_f32(hi)); }
}
__global__ void ev_cvnormalize_kernel(const float* __restrict__ img, int npix, const uint32_t* __restrict__ mm, uint8_t* __restrict__ out)
{
    const double smin = (double)dec_f32(mm[0]), smax = (double)dec_f32(mm[1]);
    const double scale = 255.0 * (smax - smin > 2.2204460492503131e-16 ? 1. / (smax - smin) : 0);
    const double shift = 0.0 - smin * scale;
    const float fs = (float)scale, fh = (float)shift;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const float m = img[i] * fs;
        const float v = m + fh;
        out[i] = (uint8_t)min(max(__float2int_rn(v), 0), 255);
    }
}
__global__ void ev_mm_reset_kernel(uint32_t* mm) { mm[0] = 0xffffffffu; mm[1] = 0u; }

int ev_cvnormalize_dev(eorb_ctx* c, const float* d_img, int npix, uint32_t* d_mm, uint8_t* d_out)
{
    ProfScope ps(c, "ev_cvnormalize");
    ev_mm_reset_kernel<<<1, 1, 0, c->stream>>>(d_mm);
    ev_minmax_final_kernel<<<64, 256, 0, c->stream>>>(d_img, npix, d_mm);
    ev_cvnormalize_kernel<<<64, 256, 0, c->stream>>>(d_img, npix, d_mm, d_out);
    EORB_LAUNCH_CHECK(c, "cv::normalize kernels");
    return EORB_OK;
}

// the same for nimg images back to back (the reconstructions of one motion-compensation dispatch): blockIdx.y = image, mm[2 * image]
__global__ void ev_mm_reset_n_kernel(uint32_t* mm, int nimg) { const int i = threadIdx.x; if (i < nimg) { mm[2 * i] = 0xffffffffu; mm[2 * i + 1] = 0u; } }
__global__ void ev_minmax_final_n_kernel(const float* __restrict__ imgs, int npix, uint32_t* __restrict__ mm)
{
    // (four pixels per load and one trip per thread for a 240 x 180 image: with one pixel per trip every thread waited for six
    // dependent round trips, 17 us for the five images of a contest)
    const float* img = imgs + (size_t)blockIdx.y * npix;
    float lo = img[0], hi = img[0];
    const bool vec = (((uintptr_t)img) & 15) == 0;
    const int n4 = vec ? npix >> 2 : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const float4 q = ((const float4*)img)[i];
        lo = fminf(fminf(lo, q.x), fminf(q.y, fminf(q.z, q.w))); hi = fmaxf(fmaxf(hi, q.x), fmaxf(q.y, fmaxf(q.z, q.w)));
    }
    for (int i = 4 * n4 + blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) { lo = fminf(lo, img[i]); hi = fmaxf(hi, img[i]); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo = fminf(lo, __shfl_xor(lo, d, 64)); hi = fmaxf(hi, __shfl_xor(hi, d, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[2 * blockIdx.y], enc_f32(lo)); atomicMax(&mm[2 * blockIdx.y + 1], enc_f32(hi)); }
}
__global__ void ev_cvnormalize_n_kernel(const float* __restrict__ imgs, int npix, const uint32_t* __restrict__ mm, uint8_t* __restrict__ outs)
{
    const float* img = imgs + (size_t)blockIdx.y * npix; uint8_t* out = outs + (size_t)blockIdx.y * npix;
    const double smin = (double)dec_f32(mm[2 * blockIdx.y]), smax = (double)dec_f32(mm[2 * blockIdx.y + 1]);
    const double scale = 255.0 * (smax - smin > 2.2204460492503131e-16 ? 1. / (smax - smin) : 0);
    const double shift = 0.0 - smin * scale;
    const float fs = (float)scale, fh = (float)shift;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const float m = img[i] * fs;
        const float v = m + fh;
        out[i] = (uint8_t)min(max(__float2int_rn(v), 0), 255);
    }
}
int ev_cvnormalize_n_dev(eorb_ctx* c, const float* d_imgs, int nimg, int npix, uint32_t* d_mm, uint8_t* d_outs)
{
    ProfScope ps(c, "ev_cvnormalize");
    ev_mm_reset_n_kernel<<<1, 64, 0, c->stream>>>(d_mm, nimg);
    ev_minmax_final_n_kernel<<<dim3((unsigned)std::min(64, std::max(1, (npix / 4 + 255) / 256)), nimg), 256, 0, c->stream>>>(d_imgs, npix, d_mm);
    ev_cvnormalize_n_kernel<<<dim3(32, nimg), 256, 0, c->stream>>>(d_imgs, npix, d_mm, d_outs);
    EORB_LAUNCH_CHECK(c, "cv::normalize kernels");
    return EORB_OK;
}

// generateMCImage's decision (EvImBuilder.cpp:1205-1216): the reconstruction with the largest focus wins, the first of equals in
// insertion order (MciInfo = std::multimap<float, ..., std::greater<float>>, include/Utils/Visualization.h:29); a winning event
// histogram is replaced by the histogram of the later half of the window (image `half_img`).  focus[m] = measureImageFocus of
// method m's image (image index img_of[m]) or -1 when the method is absent.  Writes the winner's method to *winner and its u8 image to out.
struct ContestSel { int img_of[4]; int eh_method, half_img; };
__global__ void ev_contest_select_kernel(const float* __restrict__ focus_img, ContestSel S, const uint8_t* __restrict__ u8s, int npix,
                                         float* __restrict__ focus_out /* 5: per method, then the later-half histogram's */, int* __restrict__ winner, uint8_t* __restrict__ out)
{
    float best = 0.f; int w = -1;
    for (int m = 0; m < 4; m++) {
        if (S.img_of[m] < 0) continue;
        const float f = focus_img[S.img_of[m]];
        if (w < 0 || f > best) { best = f; w = m; }
    }
    const int src = (w == S.eh_method) ? S.half_img : S.img_of[w];
    const uint8_t* im = u8s + (size_t)src * npix;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) out[i] = im[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int m = 0; m < 4; m++) focus_out[m] = S.img_of[m] >= 0 ? focus_img[S.img_of[m]] : -1.0f;
        focus_out[4] = focus_img[S.half_img];
        *winner = w;
    }
}
int ev_contest_select_dev(eorb_ctx* c, const float* d_focus_img, const int img_of[4], int half_img, const uint8_t* d_u8s, int npix,
                          float* d_focus_out, int* d_winner, uint8_t* d_out)
{
    ContestSel S; for (int m = 0; m < 4; m++) S.img_of[m] = img_of[m];
    S.eh_method = 2; S.half_img = half_img;
    ev_contest_select_kernel<<<16, 256, 0, c->stream>>>(d_focus_img, S, d_u8s, npix, d_focus_out, d_winner, d_out);
    EORB_LAUNCH_CHECK(c, "ev_contest_select_kernel");
    return EORB_OK;
}

// cv::KeyPoint records -> their points (the reference points of the LK tracker, ELK_Tracker::setRefImage KLT_Tracker.cpp:36-44)
__global__ void ev_kp_points_kernel(const eorb_keypoint* __restrict__ kps, const int32_t* __restrict__ n, int cap, float* __restrict__ pts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap && i < *n) { pts[2 * i] = kps[i].x; pts[2 * i + 1] = kps[i].y; }
}
int ev_kp_points_dev(eorb_ctx* c, const eorb_keypoint* d_kps, const int32_t* d_n, int cap, float* d_pts)
{
    ev_kp_points_kernel<<<(cap + 255) / 256, 256, 0, c->stream>>>(d_kps, d_n, cap, d_pts);
    EORB_LAUNCH_CHECK(c, "ev_kp_points_kernel");
    return EORB_OK;
}

// ---- raw sensor events: tables derived from the undistortion maps (MyCalibrator::mUndistMapX/Y, Utils/MyCalibrator.cpp:164-180) ----
// per sensor pixel: integer image position floor(x) (breakFloatCoords :51-57) or round (roundFloatCoord :46-49); -32768 when the
// undistorted point fails MyCalibrator::isInImage (:31-34) and the loader would have dropped the event (EventLoader.cpp:295-296)
__global__ void ev_src_info_kernel(const float2* __restrict__ lut, int n, int W, int H, int check, int mode_count, uint32_t* __restrict__ info,
                                   int stride2 = 1 /* float2 per entry: 1 = the maps, 2 = eorb_event16 records (their x, y lead) */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 q = lut[(size_t)i * stride2];
    const bool in = (q.x >= 0 && q.x < (float)W) && (q.y >= 0 && q.y < (float)H);
    int xi = -32768, yi = -32768;
    if ((in || !check) && q.x == q.x && q.y == q.y) {
        const float fx = mode_count ? roundf(q.x) : floorf(q.x), fy = mode_count ? roundf(q.y) : floorf(q.y);
        xi = (int)fminf(fmaxf(fx, -32000.f), 32000.f); yi = (int)fminf(fmaxf(fy, -32000.f), 32000.f);
    }
    info[i] = (uint32_t)(xi & 0xffff) | ((uint32_t)(yi & 0xffff) << 16);
}

// stamp table: S[src][i][j] = exp_XY2f(i - h - xRes, j - h - yRes) (:59-65, :236-249), evaluated exactly as K2's value waves do
__global__ void ev_stamp_kernel(const float2* __restrict__ lut, const uint32_t* __restrict__ info, int n, GatherParams P,
                                float* __restrict__ stamps, int stride2 = 1, int info_stride = 1, int info_off = 0)
{
    __shared__ uint64_t tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = kExp2Tab[threadIdx.x];
    __syncthreads();
    const int SW = 2 * P.h + 1, SWP = P.stamp_colstride;
    const size_t total = (size_t)n * SW * SWP;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
        const int src = (int)(k / (SW * SWP)), r = (int)(k - (size_t)src * SW * SWP);
        const int i = r / SWP, j = r - i * SWP;
        const uint32_t w = info[(size_t)src * info_stride + info_off];
        const int xi = (int)(int16_t)(w & 0xffff), yi = (int)(int16_t)(w >> 16);
        if (xi == -32768) continue;                   // a dropped pixel has no entries: its rows are never selected (K2r's reads past a
                                                      // column's ends may touch them, but only under a zero row mask)
        float v = 0.f;
        if (j < SW) {
            const float2 q = lut[(size_t)src * stride2];
            const float xr = q.x - (float)xi, yr = q.y - (float)yi;
            const float fx = (float)(i - P.h) - xr, fy = (float)(j - P.h) - yr;
            const float xx = fx * fx, yy = fy * fy;
            float dd = xx + yy;
            dd = dd / P.two_sig2;
            v = dev_expf_nonpos<true>(-dd, tab) / P.norm;
        }
        stamps[k] = v;
    }
}

// EventDataStore::getEventChunkRectified (EventLoader.cpp:264-305) after parsing: map lookup, ts / tsFactor, checkInImage,
// order-preserving compaction (blocks of 1024 events: count -> scan over blocks -> write)
__device__ __forceinline__ bool ev_keep(const float2 q, int W, int H, int check)
{
    return !check || ((q.x >= 0 && q.x < (float)W) && (q.y >= 0 && q.y < (float)H));
}
__global__ __launch_bounds__(1024) void ev_undistort_count_kernel(const eorb_raw_event* __restrict__ raw, size_t n, const float2* __restrict__ lut,
                                                                  int LW, int W, int H, int check, uint32_t* __restrict__ blk)
{
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const size_t k = (size_t)blockIdx.x * 1024 + threadIdx.x;
    bool keep = false;
    if (k < n) keep = ev_keep(lut[(size_t)raw[k].y * LW + raw[k].x], W, H, check);
    const uint64_t m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = cnt;
}
__global__ void ev_undistort_scan_kernel(uint32_t* blk, int nblk)
{   // single thread: nblk <= a few thousand
    uint32_t run = 0;
    for (int i = 0; i < nblk; i++) { const uint32_t v = blk[i]; blk[i] = run; run += v; }
    blk[nblk] = run;
}
__global__ __launch_bounds__(1024) void ev_undistort_write_kernel(const eorb_raw_event* __restrict__ raw, size_t n, const float2* __restrict__ lut,
                                                                  int LW, int W, int H, int check, double tsFactor,
                                                                  const uint32_t* __restrict__ blk, eorb_event* __restrict__ out)
{
    __shared__ uint32_t wbase[17];
    const size_t k = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool keep = false; float2 q = make_float2(0.f, 0.f); eorb_raw_event r{};
    if (k < n) { r = raw[k]; q = lut[(size_t)r.y * LW + r.x]; keep = ev_keep(q, W, H, check); }
    const uint64_t m = __ballot(keep);
    if (lane == 0) wbase[wave + 1] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { wbase[0] = 0; for (int w = 1; w <= 16; w++) wbase[w] += wbase[w - 1]; }
    __syncthreads();
    if (keep) {
        const uint32_t pos = blk[blockIdx.x] + wbase[wave] + (uint32_t)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        eorb_event e{};
        e.ts = r.t / tsFactor; e.x = q.x; e.y = q.y; e.p = r.p ? 1 : 0;
        out[pos] = e;
    }
}

int ev_undistort_dev(eorb_ctx* c, const eorb_raw_event* d_raw, size_t n, int W, int H, double tsFactor, eorb_event* d_out, uint32_t* d_blk)
{
    const int nblk = (int)((n + 1023) / 1024);
    ProfScope ps(c, "ev_undistort");
    ev_undistort_count_kernel<<<nblk, 1024, 0, c->stream>>>(d_raw, n, (const float2*)c->lut.p, c->lut_w, W, H, c->lut_check, d_blk);
    ev_undistort_scan_kernel<<<1, 1, 0, c->stream>>>(d_blk, nblk);
    ev_undistort_write_kernel<<<nblk, 1024, 0, c->stream>>>(d_raw, n, (const float2*)c->lut.p, c->lut_w, W, H, c->lut_check, tsFactor, d_blk, d_out);
    EORB_LAUNCH_CHECK(c, "ev_undistort kernels");
    return EORB_OK;
}

// ---- text half of the event loader (EventLoader.cpp:80-92, :264-305; BaseLoader::isComment DataStore.cpp:111-114) -----------
// "ts x y p" lines -> eorb_raw_event.  K-a counts line ends per 1024-byte block, K-b scans, K-c writes the line-end positions in
// order, K-d parses one line per thread (status 0 = event, 1 = comment / blank, 2 = outside the grammar), then the kept events are
// compacted in order with the same count -> scan -> write scheme.
__device__ __forceinline__ bool txt_is_end(const char* t, size_t nbytes, size_t i)
{   // a line ends at '\n' or at the last byte of a buffer that does not end in '\n'
    return t[i] == '\n' || (i + 1 == nbytes);
}
__global__ __launch_bounds__(1024) void txt_count_kernel(const char* __restrict__ t, size_t nbytes, uint32_t* __restrict__ blk)
{
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const bool e = i < nbytes && txt_is_end(t, nbytes, i);
    const uint64_t m = __ballot(e);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = cnt;
}
__global__ __launch_bounds__(1024) void txt_lineend_kernel(const char* __restrict__ t, size_t nbytes, const uint32_t* __restrict__ blk,
                                                           uint64_t* __restrict__ lineend)
{
    __shared__ uint32_t wbase[17];
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool e = i < nbytes && txt_is_end(t, nbytes, i);
    const uint64_t m = __ballot(e);
    if (lane == 0) wbase[wave + 1] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { wbase[0] = 0; for (int w = 1; w <= 16; w++) wbase[w] += wbase[w - 1]; }
    __syncthreads();
    if (e) lineend[blk[blockIdx.x] + wbase[wave] + (uint32_t)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = i;
}

// one decimal token: digits [ '.' digits ].  Returns false when malformed.  mant = the digits without leading zeros (up to 19
// significant), frac = number of digits after the point, big = more than 19 significant digits.
__device__ __forceinline__ bool txt_number(const char* t, size_t& p, size_t end, uint64_t& mant, int& frac, bool& big)
{
    mant = 0; frac = 0; big = false;
    int nd = 0, sig = 0; bool point = false;
    while (p < end) {
        const char ch = t[p];
        if (ch >= '0' && ch <= '9') {
            if (mant != 0 || ch != '0') { if (sig < 19) { mant = mant * 10 + (uint64_t)(ch - '0'); sig++; } else big = true; }
            nd++; if (point) frac++;
        } else if (ch == '.' && !point) point = true;
        else break;
        p++;
    }
    return nd > 0;
}
__device__ __forceinline__ void txt_blanks(const char* t, size_t& p, size_t end) { while (p < end && (t[p] == ' ' || t[p] == '\t')) p++; }

__global__ void txt_parse_kernel(const char* __restrict__ t, size_t nbytes, const uint64_t* __restrict__ lineend, uint32_t nlines,
                                 eorb_raw_event* __restrict__ ev, uint8_t* __restrict__ status)
{
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= nlines) return;
    size_t p = li ? (size_t)lineend[li - 1] + 1 : 0;
    size_t end = (size_t)lineend[li];
    if (t[end] != '\n') end++;                                    // last line without a newline: its last byte belongs to it
    if (end > p && t[end - 1] == '\r') end--;
    txt_blanks(t, p, end);
    uint8_t st = 1;                                               // comment or blank
    eorb_raw_event e{};
    if (p < end && t[p] != '#') {
        st = 2;
        uint64_t m[4]; int fr[4]; bool big[4]; bool ok = true;
        for (int k = 0; k < 4 && ok; k++) { ok = txt_number(t, p, end, m[k], fr[k], big[k]); txt_blanks(t, p, end); }
        ok = ok && p == end;
        // ts: Clinger's exact case -- integer mantissa < 2^53 and 10^frac exact in double: one correctly rounded division
        if (ok && (big[0] || m[0] >= (1ull << 53) || fr[0] > 22)) ok = false;
        // x, y: integer-valued, 0..65535;  p: 0 or 1
        for (int k = 1; k <= 2 && ok; k++) {
            uint64_t v = m[k];
            for (int d = 0; d < fr[k] && ok; d++) { if (v % 10) ok = false; v /= 10; }
            if (ok && (big[k] || v > 65535)) ok = false;
            m[k] = v;
        }
        if (ok && (big[3] || fr[3] != 0 || m[3] > 1)) ok = false;
        if (ok) {
            double p10 = 1.0;
            for (int d = 0; d < fr[0]; d++) p10 *= 10.0;          // exact up to 1e22
            e.t = (double)m[0] / p10;
            e.x = (uint16_t)m[1]; e.y = (uint16_t)m[2]; e.p = (uint32_t)m[3];
            st = 0;
        }
    }
    ev[li] = e; status[li] = st;
}

__global__ __launch_bounds__(1024) void txt_keep_count_kernel(const uint8_t* __restrict__ status, uint32_t nlines, uint32_t* __restrict__ blk,
                                                              uint32_t* __restrict__ first_bad)
{
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint8_t st = i < nlines ? status[i] : 1;
    if (st == 2) atomicMin(first_bad, i);
    const uint64_t m = __ballot(st == 0);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = cnt;
}
__global__ __launch_bounds__(1024) void txt_keep_write_kernel(const uint8_t* __restrict__ status, const eorb_raw_event* __restrict__ ev,
                                                              uint32_t nlines, const uint32_t* __restrict__ blk, eorb_raw_event* __restrict__ out)
{
    __shared__ uint32_t wbase[17];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool keep = i < nlines && status[i] == 0;
    const uint64_t m = __ballot(keep);
    if (lane == 0) wbase[wave + 1] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { wbase[0] = 0; for (int w = 1; w <= 16; w++) wbase[w] += wbase[w - 1]; }
    __syncthreads();
    if (keep) out[blk[blockIdx.x] + wbase[wave] + (uint32_t)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = ev[i];
}
__global__ void txt_set_u32_kernel(uint32_t* p, uint32_t v) { *p = v; }

// d_text: nbytes of text; d_lineend / d_ev / d_status / d_out sized for max_lines; d_blk: (nbytes + 1023) / 1024 + 4 words.
// h_res[0] = lines, h_res[1] = events kept, h_res[2] = first malformed line (0xffffffff = none)
int ev_parse_text_dev(eorb_ctx* c, const char* d_text, size_t nbytes, uint64_t* d_lineend, eorb_raw_event* d_ev, uint8_t* d_status,
                      eorb_raw_event* d_out, uint32_t* d_blk, size_t max_lines, uint32_t h_res[3])
{
    const int nblk = (int)((nbytes + 1023) / 1024);
    ProfScope ps(c, "ev_parse_text");
    txt_count_kernel<<<nblk, 1024, 0, c->stream>>>(d_text, nbytes, d_blk);
    ev_undistort_scan_kernel<<<1, 1, 0, c->stream>>>(d_blk, nblk);
    uint32_t nlines = 0;
    EORB_HIP(c, hipMemcpyAsync(&nlines, d_blk + nblk, 4, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, hipStreamSynchronize(c->stream));
    h_res[0] = nlines; h_res[1] = 0; h_res[2] = 0xffffffffu;
    if (nlines == 0) return EORB_OK;
    if (nlines > max_lines) return set_err(c, EORB_E_CAPACITY, "parse_events_text: %u lines, room for %zu", nlines, max_lines);
    txt_lineend_kernel<<<nblk, 1024, 0, c->stream>>>(d_text, nbytes, d_blk, d_lineend);
    txt_parse_kernel<<<(nlines + 255) / 256, 256, 0, c->stream>>>(d_text, nbytes, d_lineend, nlines, d_ev, d_status);
    const int nb2 = (int)((nlines + 1023) / 1024);
    uint32_t* d_blk2 = d_blk;                                      // the line-end block sums are no longer needed
    uint32_t* d_bad = d_blk + nb2 + 2;
    txt_set_u32_kernel<<<1, 1, 0, c->stream>>>(d_bad, 0xffffffffu);
    txt_keep_count_kernel<<<nb2, 1024, 0, c->stream>>>(d_status, nlines, d_blk2, d_bad);
    ev_undistort_scan_kernel<<<1, 1, 0, c->stream>>>(d_blk2, nb2);
    txt_keep_write_kernel<<<nb2, 1024, 0, c->stream>>>(d_status, d_ev, nlines, d_blk2, d_out);
    EORB_LAUNCH_CHECK(c, "ev_parse_text kernels");
    uint32_t two[2] = {0, 0};
    EORB_HIP(c, hipMemcpyAsync(&two[0], d_blk2 + nb2, 4, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, hipMemcpyAsync(&two[1], d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, hipStreamSynchronize(c->stream));
    h_res[1] = two[0]; h_res[2] = two[1];
    return EORB_OK;
}

// ---- float events in bulk -------------------------------------------------------------------------------------------------
// The reference hands ev2im_gauss float coordinates, but a loader produces them from its undistortion maps: a batch of millions of
// events takes only as many distinct (x, y) as the sensor has pixels.  Per call the distinct positions are collected in a hash table
// (key = the two floats' bit patterns = a float2: the table IS the "undistortion map" of the call, slot = "sensor pixel", empty
// slots = (NaN, NaN) = dropped), whose slots become the rows of a stamp table exactly as the maps of the raw path do
// (ev_src_info_kernel / ev_stamp_kernel evaluate the same arithmetic K2's value waves do); the events become 4-byte records naming
// their slot, and the raw path (K2r: table reads instead of 49 f64 expf per event) does the rest.  Falls back to K2 when the
// positions do not repeat (motion-compensated events).
constexpr int kDdLog = 20;                                         // table slots = 2^20; at most half may fill
constexpr uint64_t kDdEmpty = ~0ull;                               // (NaN, NaN): never a position that is looked up (NaN events are dropped)
__device__ __forceinline__ uint32_t dd_hash(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}
// every event: the slot of its position (inserted if new) | negative polarity << 31 -> rec[k]
__global__ __launch_bounds__(256) void dd_insert_kernel(const eorb_event16* __restrict__ ev, int64_t n, unsigned long long* __restrict__ tab, int* __restrict__ cnt,
                                                        uint32_t* __restrict__ rec)
{
    const uint32_t mask = (1u << kDdLog) - 1u;
    constexpr int U = 4;                                              // events per thread and round: loads and first probes in flight together
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k0 < n; k0 += stride * U) {
        uint4 q[U]; unsigned long long first[U]; uint32_t h0[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int64_t k = k0 + u * stride; q[u] = k < n ? *(const uint4*)&ev[k] : make_uint4(0x7fc00000u, 0x7fc00000u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < U; u++) { h0[u] = dd_hash((uint64_t)q[u].x | ((uint64_t)q[u].y << 32)) & mask; first[u] = tab[h0[u]]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t k = k0 + u * stride;
            if (k >= n) continue;
            const float x = __uint_as_float(q[u].x), y = __uint_as_float(q[u].y);
            const uint32_t neg = q[u].w & 0x80000000u;                 // sign bit of t
            if (!(x == x && y == y)) { rec[k] = kHashDropped | neg; continue; }        // never in the image: no position needed
            const unsigned long long key = (unsigned long long)q[u].x | ((unsigned long long)q[u].y << 32);
            uint32_t h = h0[u];
            unsigned long long cur = first[u];                         // (a stale EMPTY only costs the compare-and-swap below)
            int probes = 0;
            for (;;) {
                if (cur == key) break;
                if (cur == kDdEmpty) {
                    cur = atomicCAS(&tab[h], kDdEmpty, key);
                    if (cur == kDdEmpty) { atomicAdd(&cnt[0], 1); break; }
                    if (cur == key) break;
                }
                h = (h + 1) & mask;
                if (++probes > 256) { atomicOr(&cnt[1], 1); break; }   // table crowded: the caller falls back
                cur = tab[h];
            }
            rec[k] = h | neg;
        }
    }
}

// arithmetic constants of the Gaussian stamp (shared by all gather kernels)
static GatherParams ev_gather_params(int W, int H, int h, int TX, int TY, int NT, int mode_count, int nb, float sigma)
{
    const float sig2 = sigma * sigma;
    GatherParams G{W, H, h, TX, TY, NT, mode_count, nb, 2.0f * sig2,
                   2.0f * (float)3.1415926535897932384626433832795 * sig2, 0.f, 0.f, 0, 0, nullptr, 0, 0};
    {
        int ex2 = 0;
        const float mant = frexpf(G.two_sig2, &ex2);
        G.div_is_pow2 = (mant == 0.5f) && ex2 > -100 && ex2 < 100;
        G.inv_two_sig2 = G.div_is_pow2 ? 1.0f / G.two_sig2 : 0.f;
        G.rcp_norm = (float)(1.0 / (double)G.norm);
        // the residual ev*2^-24 must stay a normal float: exp(-dd_max) > 1e-27, dd_max = (h+1)^2 / sig2
        const double ddmax = (double)(h + 1) * (h + 1) / (double)sig2;
        G.fast_norm = (ddmax < 60.0) && (G.norm < 1e3f) && (G.norm > 1e-3f);
    }
    return G;
}

// raw events: the tables derived from the maps (integer position of every sensor pixel, its stamp); rebuilt only when (image size,
// sigma, mode) change
static int ev_raw_tables(eorb_ctx* c, int W, int H, int h, float sigma, int mode_count, GatherParams& G, bool hashed = false, bool want_slots = false)
{
    int rc;
    // tables derived from the maps; rebuilt only when (image size, sigma, mode) change
    const int nsrc = c->lut_w * c->lut_h;
    const int SW = 2 * h + 1, SWP = (SW + 3) & ~3;
    G.stamp_stride = SW * SWP; G.stamp_colstride = SWP;
    if (c->lut_key_W != W || c->lut_key_H != H || c->lut_key_sigma != sigma || c->lut_key_mode != mode_count) {
        ProfScope ps(c, "ev_stamp_tables");
        const int TXs = (W + kTile - 1) / kTile, TYs = (H + kTile - 1) / kTile;
        if (!(hashed && c->dd_src_info_done)) {
            if ((rc = ensure(c, c->src_info, sizeof(uint32_t) * (size_t)nsrc))) return rc;
            ev_src_info_kernel<<<(nsrc + 255) / 256, 256, 0, c->stream>>>((const float2*)c->lut.p, nsrc, W, H, c->lut_check, mode_count,
                                                                            (uint32_t*)c->src_info.p);
            EORB_LAUNCH_CHECK(c, "ev_src_info_kernel");
            c->sl_launched = 0;
        }
        c->dd_src_info_done = 0;
        if (!(hashed && want_slots)) c->sl_launched = 0;      // (an assignment launched ahead by the float bulk path serves that call only)
        c->sl_ok = 0;
        bool need_stamps = !mode_count;
        if (!mode_count && hashed && want_slots) {
            // the per-call positions of float events (2^20 table rows, most of them empty): the slot form computes its rows from the
            // positions, so the 235 MB stamp table is only built when that form cannot serve the call
            if ((rc = ev_slots_prepare(c, W, H, h, TXs, TYs, nullptr, G.stamp_stride, G.stamp_colstride, G.two_sig2, G.norm))) return rc;
            if (c->sl_ok) need_stamps = false;
        }
        if (need_stamps) {
            if ((size_t)nsrc * SW * SWP * 4 + 256 >= ((size_t)1 << 32)) return set_err(c, EORB_E_CAPACITY, "ev_accumulate: stamp table of %d sensor pixels x %d taps is too large", nsrc, SW * SWP);
            if ((rc = ensure(c, c->stamps, sizeof(float) * ((size_t)nsrc * SW * SWP + 2 * kStampPad)))) return rc;
            EORB_HIP(c, hipMemsetAsync(c->stamps.p, 0, sizeof(float) * kStampPad, c->stream));
            EORB_HIP(c, hipMemsetAsync((float*)c->stamps.p + kStampPad + (size_t)nsrc * SW * SWP, 0, sizeof(float) * kStampPad, c->stream));
            ev_stamp_kernel<<<2048, 256, 0, c->stream>>>((const float2*)c->lut.p, (const uint32_t*)c->src_info.p, nsrc, G,
                                                         (float*)c->stamps.p + kStampPad);
            EORB_LAUNCH_CHECK(c, "ev_stamp_kernel");
            if (!hashed && (rc = ev_slots_prepare(c, W, H, h, TXs, TYs, (const float*)c->stamps.p + kStampPad, G.stamp_stride, G.stamp_colstride, G.two_sig2, G.norm))) return rc;
        }
        c->lut_key_W = W; c->lut_key_H = H; c->lut_key_sigma = sigma; c->lut_key_mode = mode_count;
    }
    G.stamps = (const float*)c->stamps.p + kStampPad;     // K2r reads up to 7 floats before / behind a column
    return EORB_OK;
}

// eorb_raw_event4 (x | p << 15 | y << 16: the 4-byte wire record of a sensor event for the images, which never read the time
// stamp) -> eorb_raw_event, for the accumulation forms that read 16-byte records
__global__ void ev_unpack4_kernel(const uint32_t* __restrict__ in, int64_t n, eorb_raw_event* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t q = in[i];
        eorb_raw_event e; e.x = (uint16_t)(q & 0x7fffu); e.y = (uint16_t)(q >> 16); e.p = (q >> 15) & 1u; e.t = 0.0;
        out[i] = e;
    }
}

// K2d on its own: up to 8 slices given by their event ranges [beg, end) inside one array (the ranges may overlap), float or raw events,
// Gaussian stamp.  ev_accumulate_dev takes this path for the live per-slice calls; the motion-compensation contest
// (eorb_ev_mc_contest) hands it the reconstructions of one window in one launch.
int ev_direct_slices_dev(eorb_ctx* c, const void* d_events, int raw, const int64_t* beg, const int64_t* end, int B, int W, int H,
                         float sigma, int pol, float* d_f32, uint8_t* d_u8, int normalized, uint32_t* d_minmax_enc, bool mm_preset)
{
    if (B < 1 || B > kDirectSlices) return set_err(c, EORB_E_ARG, "direct slices: %d slices", B);
    const int h = (int)ceil((double)sigma * 3.0);
    if (h > 8) return set_err(c, EORB_E_CONFIG, "ev_accumulate: sigma %.3f gives half window %d > 8", sigma, h);
    const int R = (2 * h <= kTile) ? ((h == 0) ? 1 : 2) : 3;
    const int TX = (W + kTile - 1) / kTile, TY = (H + kTile - 1) / kTile, NT = TX * TY;
    int nbits = 1; while ((1 << nbits) < NT) nbits++;
    const int dup = R * R;
    DirectSlices S{};
    for (int b = 0; b < B; b++) {
        if (end[b] < beg[b]) return set_err(c, EORB_E_ARG, "ev_accumulate: offsets not monotone");
        if ((end[b] - beg[b]) * dup >= (int64_t)1 << 31) return set_err(c, EORB_E_CAPACITY, "ev_accumulate: %lld events in one slice", (long long)(end[b] - beg[b]));
        S.beg[b] = beg[b]; S.end[b] = end[b];
    }
    int rc;
    GatherParams G = ev_gather_params(W, H, h, TX, TY, NT, 0, B * NT, sigma);
    if (raw && (rc = ev_raw_tables(c, W, H, h, sigma, 0, G))) return rc;
    // every event of the call resolved once into its list entry (ev_pre_kernel), over the span of the slices
    int64_t lo = S.beg[0], hi = S.end[0];
    for (int b = 1; b < B; b++) { lo = std::min(lo, S.beg[b]); hi = std::max(hi, S.end[b]); }
    const int64_t span = std::max<int64_t>(hi - lo, 0);
    if (span >= (int64_t)1 << 30) return set_err(c, EORB_E_CAPACITY, "ev_accumulate: %lld events in one direct call", (long long)span);
    if ((rc = ensure(c, c->ev_info, sizeof(uint2) * (size_t)std::max<int64_t>(span, 1)))) return rc;
    S.tab_base = lo;
    const bool mm_in_pre = !mm_preset && span > 0;               // (ev_pre_kernel's first threads initialise the running extremes)
    uint2* d_pre = (uint2*)c->ev_info.p;
    const eorb_raw_event* d_first = (const eorb_raw_event*)d_events + lo;
    if (!raw) {
        // float events: the taps per EVENT (row = the event's index in the span), as the maps' table holds them per sensor pixel
        const int SW = 2 * h + 1, SWP = (SW + 3) & ~3;
        if (span * SW * SWP * 4 >= (int64_t)1 << 32) return set_err(c, EORB_E_CAPACITY, "ev_accumulate: %lld float events x %d taps exceed the per-event stamp table", (long long)span, SW * SWP);
        G.stamp_stride = SW * SWP; G.stamp_colstride = SWP;
        if ((rc = ensure(c, c->ev_stamps, sizeof(float) * (size_t)std::max<int64_t>(span, 1) * SW * SWP))) return rc;
        if (span) {
            ProfScope ps(c, "ev_stamp_tables");
            ev_pre_kernel<true><<<(int)((span + 255) / 256), 256, 0, c->stream>>>(d_first, (int)span, W, H, 0, 0, nullptr, 0u, d_pre, mm_in_pre ? d_minmax_enc : nullptr, B);
            ev_stamp_kernel<<<(int)std::min<int64_t>((span * SW * SWP + 255) / 256, 4096), 256, 0, c->stream>>>((const float2*)d_first, (const uint32_t*)d_pre, (int)span, G,
                                                                                                          (float*)c->ev_stamps.p, 2, 2, 1);
            EORB_LAUNCH_CHECK(c, "per-event tables");
        }
        G.stamps = (const float*)c->ev_stamps.p;
    } else if (span) {
        ProfScope ps(c, "ev_stamp_tables");
        ev_pre_kernel<false><<<(int)((span + 255) / 256), 256, 0, c->stream>>>(d_first, (int)span, W, H, c->lut_w, c->lut_h, (const uint32_t*)c->src_info.p, 0u, d_pre, mm_in_pre ? d_minmax_enc : nullptr, B);
        EORB_LAUNCH_CHECK(c, "ev_pre_kernel");
    }
    if (!mm_preset && !mm_in_pre) {
        ProfScope ps(c, "ev_minmax_init");
        ev_minmax_init_kernel<<<(B + 63) / 64, 64, 0, c->stream>>>(d_minmax_enc, B);
    }
    {
        ProfScope ps(c, "ev_gather");
        if (pol) ev_gather_direct_kernel<true><<<B * NT, 64, 0, c->stream>>>(d_pre, S, G, d_f32, d_minmax_enc);
        else ev_gather_direct_kernel<false><<<B * NT, 64, 0, c->stream>>>(d_pre, S, G, d_f32, d_minmax_enc);
        EORB_LAUNCH_CHECK(c, "ev_gather_direct_kernel");
    }
    if (normalized && d_u8) {
        ProfScope ps(c, "ev_normalize");
        dim3 grid((W * H + 255) / 256 > 64 ? 64 : (W * H + 255) / 256, B);
        ev_normalize_kernel<<<grid, 256, 0, c->stream>>>(d_f32, d_minmax_enc, d_u8, W * H, 0);
        EORB_LAUNCH_CHECK(c, "ev_normalize_kernel");
    }
    return EORB_OK;
}

// eorb_raw_event2 (the sensor pixel's linear index y * LW + x; 0xffff = no event: the 2-byte wire record of polarity-free images on
// sensors of at most 65 535 pixels) -> eorb_raw_event
__global__ void ev_unpack2_kernel(const uint16_t* __restrict__ in, int64_t n, int LW, eorb_raw_event* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t q = in[i];
        eorb_raw_event e; e.x = (uint16_t)(q == 0xffffu ? 0xffffu : q % (uint32_t)LW); e.y = (uint16_t)(q == 0xffffu ? 0xffffu : q / (uint32_t)LW); e.p = 1u; e.t = 0.0;
        out[i] = e;
    }
}

// ---- float events in bulk, across calls: the position dictionary -----------------------------------------------------------------
// A loader's maps do not change between calls, so the positions of one call's events are those of the next.  After a call whose
// positions were tabulated (dd_insert_kernel) and number at most 65 535, they are frozen into a DICTIONARY: a compact open-addressing
// table { x bits, y bits, dense id } (2^17 entries: L2-resident) and, per dense id, what the maps' tables hold per sensor pixel (position,
// integer position, slot tables, rows).  The following calls run the slot form straight on the float events: its count pass looks
// every position up (sl_count_lds_kernel<16, true>) and writes the dense id as the hashed record -- one pass over the events where the
// per-call tabulation makes two, no table to clear and rebuild.  A position the dictionary does not hold is counted; the call then
// falls back to the per-call tabulation (whose table becomes the next dictionary).
constexpr int kPdLog = 17;
__global__ void pd_build_kernel(const unsigned long long* __restrict__ tab, int nslots, int max_ids, int* __restrict__ cnt, float2* __restrict__ lut,
                                uint4* __restrict__ hash)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    const unsigned long long key = tab[i];
    if (key == kDdEmpty) return;
    const int id = atomicAdd(&cnt[0], 1);
    if (id >= max_ids) return;
    const uint32_t kx = (uint32_t)key, ky = (uint32_t)(key >> 32);
    lut[id] = make_float2(__uint_as_float(kx), __uint_as_float(ky));
    const uint32_t mask = (1u << kPdLog) - 1u;
    uint32_t h = dd_hash(key) & mask;
    // claim a free entry through its id word (0xffffffff = free), then fill the key: the table is read only after this kernel
    for (;;) {
        if (atomicCAS(&hash[h].z, 0xffffffffu, (uint32_t)id) == 0xffffffffu) { hash[h].x = kx; hash[h].y = ky; break; }
        h = (h + 1) & mask;
    }
}

// the tables of the call's positions (the context's dd_* set, valid right after a per-call tabulation) -> the frozen dictionary
static int pd_freeze(eorb_ctx* c, int W, int H, int h, float sigma, int TX, int TY, int npos)
{
    c->pd_valid = 0;
    if (npos < 1 || npos > 65535) return EORB_OK;                      // (dense ids travel as hashed records below 65 536 rows: the LDS count pass)
    const size_t cap = (size_t)1 << kDdLog, hcap = (size_t)1 << kPdLog;
    int rc;
    if ((rc = ensure(c, c->pd_hash, sizeof(uint4) * hcap)) || (rc = ensure(c, c->pd_lut, sizeof(float2) * 65536)) || (rc = ensure(c, c->pd_cnt, 64))) return rc;
    EORB_HIP(c, hipMemsetAsync(c->pd_hash.p, 0xff, sizeof(uint4) * hcap, c->stream));
    EORB_HIP(c, hipMemsetAsync(c->pd_cnt.p, 0, 64, c->stream));
    pd_build_kernel<<<(int)((cap + 255) / 256), 256, 0, c->stream>>>((const unsigned long long*)c->dd_tab.p, (int)cap, 65535, (int*)c->pd_cnt.p, (float2*)c->pd_lut.p, (uint4*)c->pd_hash.p);
    EORB_LAUNCH_CHECK(c, "pd_build_kernel");
    // the per-id tables: the raw path's own builders on the dictionary's positions as a (K x 1) "sensor"
    auto swap_tables = [&]() {
        std::swap(c->lut, c->pd_lut); std::swap(c->src_info, c->pd_src_info);
        std::swap(c->sl_tab, c->pd_sl_tab); std::swap(c->sl_tile, c->pd_sl_tile); std::swap(c->sl_rows, c->pd_sl_rows);
        std::swap(c->sl_null, c->pd_sl_null); std::swap(c->sl_info_off, c->pd_sl_info_off);
    };
    swap_tables();
    const int sw = c->lut_w, sh = c->lut_h, sc = c->lut_check, sok = c->sl_ok, sl0 = c->sl_launched;
    c->lut_w = npos; c->lut_h = 1; c->lut_check = 0; c->sl_launched = 0;
    rc = ensure(c, c->src_info, sizeof(uint32_t) * (size_t)npos);
    if (!rc) {
        ev_src_info_kernel<<<(npos + 255) / 256, 256, 0, c->stream>>>((const float2*)c->lut.p, npos, W, H, 0, 0, (uint32_t*)c->src_info.p);
        const float sig2 = sigma * sigma;
        rc = ev_slots_prepare(c, W, H, h, TX, TY, nullptr, 0, 0, 2.0f * sig2, 2.0f * (float)3.1415926535897932384626433832795 * sig2);
    }
    const int ok = c->sl_ok;
    c->lut_w = sw; c->lut_h = sh; c->lut_check = sc; c->sl_ok = sok; c->sl_launched = sl0;
    swap_tables();
    if (rc) return rc;
    // the count pass keeps the dense ids' tile ranges and its per-wavefront counters in LDS (ev_slots.hip, sl_count_lds_kernel)
    const int NTp = (TX * TY + 1) & ~1;
    const bool lds_ok = TX <= 127 && TY <= 127 && 4 * (((size_t)npos + 2) / 2) + (size_t)16 * NTp * 2 <= 159 * 1024;
    if (ok && lds_ok) { c->pd_valid = 1; c->pd_K = npos; c->pd_W = W; c->pd_H = H; c->pd_sigma = sigma; }
    return EORB_OK;
}

// a call served by the dictionary; *missed = 1 when a position was not in it (the images are then incomplete: the caller redoes the call)
static int pd_accumulate(eorb_ctx* c, const eorb_event16* d_src, const int64_t* off, int B, int W, int H, int TX, int TY, float sigma,
                         float* d_f32, uint8_t* d_u8, int normalized, uint32_t* d_minmax_enc, int64_t n0, int* missed)
{
    int rc;
    if ((rc = ensure(c, c->dd_ev, 4 * (size_t)std::max<int64_t>(n0, 1)))) return rc;
    EORB_HIP(c, hipMemsetAsync(c->pd_cnt.p, 0, 64, c->stream));
    auto swap_tables = [&]() {
        std::swap(c->lut, c->pd_lut); std::swap(c->src_info, c->pd_src_info);
        std::swap(c->sl_tab, c->pd_sl_tab); std::swap(c->sl_tile, c->pd_sl_tile); std::swap(c->sl_rows, c->pd_sl_rows);
        std::swap(c->sl_null, c->pd_sl_null); std::swap(c->sl_info_off, c->pd_sl_info_off);
    };
    swap_tables();
    const int sw = c->lut_w, sh = c->lut_h, sok = c->sl_ok;
    c->lut_w = c->pd_K; c->lut_h = 1; c->sl_ok = 1;
    {
        ProfScope ps(c, "ev_minmax_init");
        ev_minmax_init_kernel<<<(B + 63) / 64, 64, 0, c->stream>>>(d_minmax_enc, B);
    }
    SlotDict D{(const uint4*)c->pd_hash.p, (1u << kPdLog) - 1u, (uint32_t*)c->dd_ev.p, (int*)c->pd_cnt.p, c->pd_K < 65535 ? 1 : 0};
    rc = ev_slots_accumulate(c, d_src, 16, off, B, W, H, TX, TY, d_f32, d_minmax_enc, &D);
    c->lut_w = sw; c->lut_h = sh; c->sl_ok = sok;
    swap_tables();
    if (rc < 0) return rc;
    if (rc > 0) { *missed = 1; return EORB_OK; }                        // the batch's shape does not fit the slot form: the per-call path decides
    if (normalized && d_u8) {
        ProfScope ps(c, "ev_normalize");
        dim3 grid((W * H + 255) / 256 > 64 ? 64 : (W * H + 255) / 256, B);
        ev_normalize_kernel<<<grid, 256, 0, c->stream>>>(d_f32, d_minmax_enc, d_u8, W * H, 0);
        EORB_LAUNCH_CHECK(c, "ev_normalize_kernel");
    }
    // the call's one wait: did every position have its entry?
    int* rb = readback_buf(c);
    if (!rb) return set_err(c, EORB_E_HIP, "pinned alloc failed");
    EORB_HIP(c, hipMemcpyAsync(rb, c->pd_cnt.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, hipStreamSynchronize(c->stream));
    *missed = rb[0] != 0;
    return EORB_OK;
}

// ---------------------------------------------------------------------------------------------------
int ev_accumulate_dev(eorb_ctx* c, const void* d_events, int raw, const int64_t* h_offsets, int B, int W, int H,
                      float sigma, int pol, int mode_count, float* d_f32, uint8_t* d_u8, int normalized,
                      uint32_t* d_minmax_enc)
{
    const bool mm_preset = c->mm_preset;                 // (only the binning-free form takes the caller's word for it; the others initialise the extremes themselves)
    c->mm_preset = false;
    if (B <= 0 || W <= 0 || H <= 0) return set_err(c, EORB_E_ARG, "ev_accumulate: bad size");
    if (raw && !c->lut_w) return set_err(c, EORB_E_NOTCONF, "ev_accumulate: raw events need eorb_set_undistort_maps first");
    const eorb_event16* d_ev = (const eorb_event16*)d_events;          // eorb_raw_event has the same 16-byte stride
    const int h = mode_count ? 0 : (int)ceil((double)sigma * 3.0);     // lenHalfWin :222
    if (h > 8) return set_err(c, EORB_E_CONFIG, "ev_accumulate: sigma %.3f gives half window %d > 8", sigma, h);
    // exp(-d^2 / 2 sigma^2) underflows to 0 inside the window below sigma ~ 0.197 (d^2 up to 8 at h = 1): with polarity the
    // reference's running maximum then depends on visits that add nothing (see the gather kernels)
    if (pol && !mode_count && sigma < 0.2f) return set_err(c, EORB_E_CONFIG, "ev_accumulate: polarity images need sigma >= 0.2 (got %.3f)", sigma);
    const int R = (2 * h <= kTile) ? ((h == 0) ? 1 : 2) : 3;            // max tiles an event spans per axis
    const int TX = (W + kTile - 1) / kTile, TY = (H + kTile - 1) / kTile, NT = TX * TY;
    int nbits = 1; while ((1 << nbits) < NT) nbits++;
    const int dup = R * R;
    const bool hashed = raw == 2;                        // (recursion of the block below: d_events are 4-byte hashed records)
    const bool packed4 = raw == 3;                       // 4-byte sensor records (eorb_raw_event4)
    const bool packed2 = raw == 4;                       // 2-byte sensor records (eorb_raw_event2: y * LW + x)
    if (packed2 && (int64_t)c->lut_w * c->lut_h > 65535) return set_err(c, EORB_E_ARG, "ev_accumulate: 2-byte records need a sensor of at most 65 535 pixels (the maps are %dx%d)", c->lut_w, c->lut_h);
    if (!raw && !mode_count && c->dbg_gather_form != 1 && h_offsets[B] - h_offsets[0] >= c->dbg_dd_min && c->dbg_dd_min > 0) {
        // float events in bulk: tabulate their distinct positions, continue on the raw path (see dd_insert_kernel)
        const int64_t n0 = h_offsets[B] - h_offsets[0];
        const int64_t ps0 = n0 / B;
        const int chunk0 = ps0 >= (int64_t)1 << 17 ? 2048 : (ps0 >= (int64_t)1 << 14 ? 1024 : 256);
        const int NTp0 = (NT + 1) & ~1;
        static const int scat_env = [] { const char* e = getenv("EORB_SCATTER"); return e ? atoi(e) : 2; }();
        // the hashed records are only read by the count kernel and the second form of the scatter
        const bool fits = scat_env != 1 && TX < 256 && TY < 256 &&
            (((size_t)chunk0 * 8 + (size_t)chunk0 * 2 + (size_t)chunk0 * R * R * 2 + (size_t)kScatWaves * NTp0 * 2 + (size_t)(NTp0 + 2) * 2 + (size_t)NT * 4 + 15) & ~(size_t)15) <= 64 * 1024;
        const size_t cap = (size_t)1 << kDdLog;
        const int max_ids = (int)(cap / 2);
        int rc;
        const bool slot_shape = !pol && !mode_count && (c->dbg_gather_form == 0 || c->dbg_gather_form == 4) && h >= 1 && h <= 4;
        if (fits && slot_shape && c->pd_valid && c->pd_W == W && c->pd_H == H && c->pd_sigma == sigma && c->pd_cooldown == 0 && c->dbg_pd != 0) {
            // the positions of an earlier call, frozen: one pass over the events (see pd_accumulate)
            std::vector<int64_t> off(B + 1);
            for (int b = 0; b <= B; b++) off[b] = h_offsets[b] - h_offsets[0];
            int missed = 0;
            if ((rc = pd_accumulate(c, (const eorb_event16*)d_events + h_offsets[0], off.data(), B, W, H, TX, TY, sigma, d_f32, d_u8, normalized, d_minmax_enc, n0, &missed))) return rc;
            if (!missed) { c->pd_hits++; return EORB_OK; }
            c->pd_misses++; c->pd_valid = 0;                            // new positions: tabulate this call's afresh (below), freeze those
        }
        if (c->pd_cooldown > 0) c->pd_cooldown--;
        if (fits) {
            if ((rc = ensure(c, c->dd_tab, 8 * cap)) || (rc = ensure(c, c->dd_cnt, 64)) || (rc = ensure(c, c->dd_ev, 4 * (size_t)n0))) return rc;
            const eorb_event16* src = (const eorb_event16*)d_events + h_offsets[0];
            int hc[2] = {0, 0};
            {
                ProfScope ps(c, "ev_dedupe");
                // the table of an earlier call is kept while the positions seen so far can still become a dictionary (a sparse call does
                // not see every sensor pixel: the table accumulates them over the calls); otherwise it starts empty
                const bool keep = c->dd_keep && slot_shape && c->dbg_pd != 0;
                if (!keep) {
                    EORB_HIP(c, hipMemsetAsync(c->dd_tab.p, 0xff, 8 * cap, c->stream));
                    EORB_HIP(c, hipMemsetAsync(c->dd_cnt.p, 0, 64, c->stream));
                } else EORB_HIP(c, hipMemsetAsync((int*)c->dd_cnt.p + 1, 0, 4, c->stream));      // (the crowding flag is per call)
                c->dd_keep = 0;
                dd_insert_kernel<<<4096, 256, 0, c->stream>>>(src, n0, (unsigned long long*)c->dd_tab.p, (int*)c->dd_cnt.p, (uint32_t*)c->dd_ev.p);
                EORB_LAUNCH_CHECK(c, "dd_insert_kernel");
                int* rb = readback_buf(c);
                if (!rb) return set_err(c, EORB_E_HIP, "pinned alloc failed");
                EORB_HIP(c, hipMemcpyAsync(rb, c->dd_cnt.p, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
                // in front of the wait: the integer positions of the table rows and the slot assignment (what the slot form's host
                // side needs read back), so that the call waits for the stream once
                if (!pol && !mode_count && (c->dbg_gather_form == 0 || c->dbg_gather_form == 2 || c->dbg_gather_form == 4) && h >= 1 && h <= 4) {
                    std::swap(c->lut, c->dd_tab); std::swap(c->src_info, c->dd_src_info);
                    std::swap(c->sl_tab, c->dd_sl_tab); std::swap(c->sl_tile, c->dd_sl_tile); std::swap(c->sl_info_off, c->dd_sl_info_off);
                    const int sw0 = c->lut_w, sh0 = c->lut_h, sok0 = c->sl_ok;
                    c->lut_w = 65536; c->lut_h = (int)(cap / 65536);
                    int rc2 = ensure(c, c->src_info, sizeof(uint32_t) * cap);
                    if (!rc2) {
                        ev_src_info_kernel<<<(int)((cap + 255) / 256), 256, 0, c->stream>>>((const float2*)c->lut.p, (int)cap, W, H, 0, mode_count, (uint32_t*)c->src_info.p);
                        rc2 = ev_slots_prepare_launch(c, W, H, h, TX, TY);
                    }
                    c->sl_ok = sok0;
                    c->lut_w = sw0; c->lut_h = sh0;
                    std::swap(c->lut, c->dd_tab); std::swap(c->src_info, c->dd_src_info);
                    std::swap(c->sl_tab, c->dd_sl_tab); std::swap(c->sl_tile, c->dd_sl_tile); std::swap(c->sl_info_off, c->dd_sl_info_off);
                    if (rc2) return rc2;
                    c->dd_src_info_done = 1;
                }
                EORB_HIP(c, hipStreamSynchronize(c->stream));
                hc[0] = rb[0]; hc[1] = rb[1];
            }
            if (!(!hc[1] && hc[0] <= max_ids)) { c->dd_src_info_done = 0; c->sl_launched = 0; }
            if (!hc[1] && hc[0] <= max_ids) {
                // the raw path on the per-call tables: the context's map state is swapped for the duration of the call
                auto swap_tables = [&]() {
                    std::swap(c->lut, c->dd_tab); std::swap(c->src_info, c->dd_src_info); std::swap(c->stamps, c->dd_stamps);
                    std::swap(c->sl_tab, c->dd_sl_tab); std::swap(c->sl_tile, c->dd_sl_tile); std::swap(c->sl_rows, c->dd_sl_rows);
                    std::swap(c->sl_ok, c->dd_sl_ok); std::swap(c->sl_null, c->dd_sl_null); std::swap(c->sl_info_off, c->dd_sl_info_off);
                };
                swap_tables();
                const int sw = c->lut_w, sh = c->lut_h, sc = c->lut_check, kW = c->lut_key_W, kH = c->lut_key_H, kM = c->lut_key_mode;
                const float kS = c->lut_key_sigma;
                c->lut_w = 65536; c->lut_h = (int)(cap / 65536); c->lut_check = 0;       // the hash table as the call's maps: slot = sensor pixel
                c->lut_key_W = c->lut_key_H = c->lut_key_mode = -1; c->lut_key_sigma = -1.f;
                std::vector<int64_t> off(B + 1);
                for (int b = 0; b <= B; b++) off[b] = h_offsets[b] - h_offsets[0];
                rc = ev_accumulate_dev(c, c->dd_ev.p, 2, off.data(), B, W, H, sigma, pol, mode_count, d_f32, d_u8, normalized, d_minmax_enc);
                swap_tables();
                c->lut_w = sw; c->lut_h = sh; c->lut_check = sc; c->lut_key_W = kW; c->lut_key_H = kH; c->lut_key_mode = kM; c->lut_key_sigma = kS;
                // these positions serve the next calls (a call that just missed the dictionary with MANY new positions -- events that are not
                // map-valued -- does not try again for a while)
                if (!rc && slot_shape && c->dbg_pd != 0) {
                    if (hc[0] <= 65535) { const int rc3 = pd_freeze(c, W, H, h, sigma, TX, TY, hc[0]); if (rc3) return rc3; c->dd_keep = 1; }
                    else c->pd_cooldown = 16;
                }
                return rc;
            }
        }
    }
    {
        // one or a few small slices of raw events (the live per-slice call): no binning at all, K2d
        const int64_t nev0 = h_offsets[B] - h_offsets[0];
        if (!hashed && !packed4 && !packed2 && !mode_count && B <= 4 && (c->dbg_gather_form == 3 || (c->dbg_gather_form == 0 && nev0 <= 16384))) {
            int64_t beg[kDirectSlices], end[kDirectSlices];
            for (int b = 0; b < B; b++) {
                if (h_offsets[b + 1] < h_offsets[b]) return set_err(c, EORB_E_ARG, "ev_accumulate: offsets not monotone");
                beg[b] = h_offsets[b]; end[b] = h_offsets[b + 1];
            }
            return ev_direct_slices_dev(c, d_events, raw, beg, end, B, W, H, sigma, pol, d_f32, d_u8, normalized, d_minmax_enc, mm_preset);
        }
    }
    // dense batches of raw events without polarity: two-byte slot lists, a tile position's rows in LDS (ev_slots.hip)
    if (raw && !mode_count && !pol && (c->dbg_gather_form == 0 || c->dbg_gather_form == 4 || (hashed && c->dbg_gather_form == 2))) {
        const int64_t nev0 = h_offsets[B] - h_offsets[0];
        const bool sparse0 = nev0 * dup < (int64_t)B * NT * 64;          // (fewer than one 64-entry batch per tile on average: K2s)
        if (c->dbg_gather_form == 4 || !sparse0) {
            int rc;
            GatherParams G = ev_gather_params(W, H, h, TX, TY, NT, mode_count, B * NT, sigma);
            if ((rc = ev_raw_tables(c, W, H, h, sigma, mode_count, G, hashed, true))) return rc;
            if (c->sl_ok) {
                {
                    ProfScope ps(c, "ev_minmax_init");
                    ev_minmax_init_kernel<<<(B + 63) / 64, 64, 0, c->stream>>>(d_minmax_enc, B);
                }
                // > 0: the batch's shape does not fit the slot form (tile grids beyond the scatter's LDS, e.g. VGA-class sensors; more than
                // 2 048 slices) and nothing was launched: the batch pipeline below serves it
                if ((rc = ev_slots_accumulate(c, d_events, hashed ? -4 : (packed4 ? 4 : (packed2 ? 2 : 16)), h_offsets, B, W, H, TX, TY, d_f32, d_minmax_enc)) < 0) return rc;
                if (rc == 0) {
                    if (normalized && d_u8) {
                        ProfScope ps(c, "ev_normalize");
                        dim3 grid((W * H + 255) / 256 > 64 ? 64 : (W * H + 255) / 256, B);
                        ev_normalize_kernel<<<grid, 256, 0, c->stream>>>(d_f32, d_minmax_enc, d_u8, W * H, mode_count);
                        EORB_LAUNCH_CHECK(c, "ev_normalize_kernel");
                    }
                    return EORB_OK;
                }
            }
        }
    }
    if (packed4 || packed2) {
        // the other forms read 16-byte records: widen the batch's events once and go on with those
        const int64_t n0 = h_offsets[B] - h_offsets[0];
        int rc;
        if ((rc = ensure(c, c->ev16, sizeof(eorb_raw_event) * (size_t)std::max<int64_t>(n0, 1)))) return rc;
        if (n0 > 0 && packed2) ev_unpack2_kernel<<<(int)std::min<int64_t>((n0 + 255) / 256, 65536), 256, 0, c->stream>>>((const uint16_t*)d_events + h_offsets[0], n0, c->lut_w, (eorb_raw_event*)c->ev16.p);
        else if (n0 > 0) ev_unpack4_kernel<<<(int)std::min<int64_t>((n0 + 255) / 256, 65536), 256, 0, c->stream>>>((const uint32_t*)d_events + h_offsets[0], n0, (eorb_raw_event*)c->ev16.p);
        EORB_LAUNCH_CHECK(c, "ev_unpack4_kernel");
        std::vector<int64_t> off(B + 1);
        for (int b = 0; b <= B; b++) off[b] = h_offsets[b] - h_offsets[0];
        return ev_accumulate_dev(c, c->ev16.p, 1, off.data(), B, W, H, sigma, pol, mode_count, d_f32, d_u8, normalized, d_minmax_enc);
    }
    // chunk list (host) -> device.  A chunk is binned by ONE wavefront, 64 events at a time: large inputs take kChunk events per
    // chunk (fewer segment tables), small ones shorter chunks so that a single 2 000-event slice is not one 32-iteration serial
    // loop (72 us on MI355X) but eight waves side by side
    const int64_t nev_all = h_offsets[B] - h_offsets[0];
    const int64_t per_slice = nev_all / B;
    static const int scat_form = [] { const char* e = getenv("EORB_SCATTER"); return e ? atoi(e) : 2; }();      // 1: first form (A/B runs)
    const int chunk_big = scat_form == 1 ? kChunk : 2048;
    const int chunk = per_slice >= (int64_t)1 << 17 ? chunk_big : (per_slice >= (int64_t)1 << 14 ? 1024 : 256);
    std::vector<ChunkDesc> cds;
    std::vector<int> slice_c0(B + 1);
    std::vector<int64_t> slice_eb(B);
    for (int b = 0; b < B; b++) {
        slice_c0[b] = (int)cds.size();
        const int64_t s = h_offsets[b], e = h_offsets[b + 1];
        if (e < s) return set_err(c, EORB_E_ARG, "ev_accumulate: offsets not monotone");
        if ((e - s) * dup >= (int64_t)1 << 31) return set_err(c, EORB_E_CAPACITY, "ev_accumulate: %lld events in one slice", (long long)(e - s));
        slice_eb[b] = (s - h_offsets[0]) * dup;                 // every event yields at most dup entries
        for (int64_t k = s; k < e; k += chunk) {
            ChunkDesc cd; cd.start = k; cd.n = (int32_t)std::min<int64_t>(chunk, e - k); cd.slice = b;
            cds.push_back(cd);
        }
    }
    slice_c0[B] = (int)cds.size();
    const int nchunks = (int)cds.size();
    const int64_t nev = h_offsets[B] - h_offsets[0];
    const size_t cd_bytes = sizeof(ChunkDesc) * (size_t)std::max(nchunks, 1);
    const size_t sc_bytes = (sizeof(int) * (size_t)(B + 1) + 7) & ~(size_t)7;
    const size_t eb_bytes = sizeof(int64_t) * (size_t)B;
    const int nb = B * NT;
    int rc;
    if ((rc = ensure(c, c->chunks, cd_bytes + sc_bytes + eb_bytes))) return rc;
    // segcnt u16 [nchunks][NT] | segbase u32 [nchunks][NT]
    const size_t cnt_bytes = (sizeof(uint16_t) * (size_t)std::max(nchunks, 1) * NT + 15) & ~(size_t)15;
    if ((rc = ensure(c, c->segoff, cnt_bytes + sizeof(uint32_t) * (size_t)std::max(nchunks, 1) * NT))) return rc;
    const int esz = pol ? 4 : 2;
    if ((rc = ensure(c, c->entries, sizeof(float) * esz * (size_t)std::max<int64_t>(nev, 1) * dup + 64))) return rc;   // + slack: K2r reads entry 0 of an empty list
    // tile_cnt | tile_base | order
    if ((rc = ensure(c, c->tile_order, sizeof(uint32_t) * 3 * (size_t)nb))) return rc;
    char* hp = (char*)pinned(c, cd_bytes + sc_bytes + eb_bytes);
    if (!hp) return set_err(c, EORB_E_HIP, "pinned alloc failed");
    if (nchunks) memcpy(hp, cds.data(), sizeof(ChunkDesc) * nchunks);
    memcpy(hp + cd_bytes, slice_c0.data(), sizeof(int) * (size_t)(B + 1));
    memcpy(hp + cd_bytes + sc_bytes, slice_eb.data(), eb_bytes);
    EORB_HIP(c, hipMemcpyAsync(c->chunks.p, hp, cd_bytes + sc_bytes + eb_bytes, hipMemcpyHostToDevice, c->stream));
    pinned_commit(c);
    const ChunkDesc* d_chunks = (const ChunkDesc*)c->chunks.p;
    const int* d_slice_c0 = (const int*)((char*)c->chunks.p + cd_bytes);
    const int64_t* d_slice_eb = (const int64_t*)((char*)c->chunks.p + cd_bytes + sc_bytes);
    uint16_t* d_segcnt = (uint16_t*)c->segoff.p;
    uint32_t* d_segbase = (uint32_t*)((char*)c->segoff.p + cnt_bytes);
    uint32_t* d_tile_cnt = (uint32_t*)c->tile_order.p;
    uint32_t* d_tile_base = d_tile_cnt + nb;
    int32_t* d_order = (int32_t*)(d_tile_cnt + 2 * (size_t)nb);

    GatherParams G = ev_gather_params(W, H, h, TX, TY, NT, mode_count, nb, sigma);
    if (raw && (rc = ev_raw_tables(c, W, H, h, sigma, mode_count, G, hashed))) return rc;
    {
        ProfScope ps(c, "ev_minmax_init");
        ev_minmax_init_kernel<<<(B + 63) / 64, 64, 0, c->stream>>>(d_minmax_enc, B);
    }
    // nearly empty tiles (fewer than one 64-entry batch per tile on average) of raw events with a Gaussian stamp: K2s, a wave per tile
    const bool sparse = raw && !mode_count && (c->dbg_gather_form == 2 || (c->dbg_gather_form == 0 && nev * dup < (int64_t)nb * 64));
    {
        BinParams P{W, H, h, TX, TY, NT, nbits, dup, mode_count, pol, raw ? 1 : 0, c->lut_w, c->lut_h, (const uint32_t*)c->src_info.p,
                    hashed ? 1 : 0};
        const size_t lds = sizeof(uint32_t) * (size_t)NT;
        if (lds > 64 * 1024) return set_err(c, EORB_E_CAPACITY, "ev_accumulate: %d tiles exceed the binning LDS", NT);
        ProfScope ps(c, "ev_bin");
        float* en = (float*)c->entries.p;
        if (nchunks) ev_count_kernel<<<nchunks, 256, lds, c->stream>>>(d_ev, d_chunks, P, d_segcnt);
        ev_scan_kernel<<<B, 1024, 0, c->stream>>>(d_slice_c0, d_segcnt, NT, d_segbase, d_tile_cnt, d_tile_base);
        if (nchunks) {
            const int NTp = (NT + 1) & ~1;
            const size_t lds2 = ((size_t)chunk * 8 + (size_t)chunk * 2 + (size_t)chunk * R * R * 2 + (size_t)kScatWaves * NTp * 2 + (size_t)(NTp + 2) * 2 + (size_t)NT * 4 + 15) & ~(size_t)15;
            // (float events with polarity carry 16-byte entries: first form)
            const bool form2 = scat_form != 1 && !(pol && !raw) && lds2 <= 64 * 1024 && TX < 256 && TY < 256;
            if (hashed && !form2) return set_err(c, EORB_E_CAPACITY, "ev_accumulate: hashed records need the second form of the scatter");
#define LAUNCH_BIN(RR, PP) do { if (form2) ev_scatter2_kernel<RR><<<nchunks, 64 * kScatWaves, lds2, c->stream>>>(d_ev, d_chunks, P, chunk, d_slice_eb, d_segbase, d_tile_base, (uint2*)en); \
                                else ev_scatter_kernel<RR, PP><<<nchunks, 64, lds, c->stream>>>(d_ev, d_chunks, P, d_slice_eb, d_segbase, d_tile_base, en); } while (0)
            const bool wide = pol && !raw;                       // raw entries keep the polarity in the sensor-pixel word
            if (R == 1) { if (wide) LAUNCH_BIN(1, true); else LAUNCH_BIN(1, false); }
            else if (R == 2) { if (wide) LAUNCH_BIN(2, true); else LAUNCH_BIN(2, false); }
            else { if (wide) LAUNCH_BIN(3, true); else LAUNCH_BIN(3, false); }
#undef LAUNCH_BIN
        }
        if (!sparse) {
            const int nblk = (nb + kOrderItems - 1) / kOrderItems;
            if ((rc = ensure(c, c->order_hist, sizeof(uint32_t) * 64 * (size_t)nblk))) return rc;
            ev_tile_hist_kernel<<<nblk, 1024, 0, c->stream>>>(d_tile_cnt, nb, (uint32_t*)c->order_hist.p);
            ev_tile_order_kernel<<<nblk, 1024, 0, c->stream>>>(d_tile_cnt, nb, (const uint32_t*)c->order_hist.p, d_order);
        }
        EORB_LAUNCH_CHECK(c, "ev_bin kernels");
    }
    {
        ProfScope ps(c, "ev_gather");
        static const int gthreads = [] { const char* e = getenv("EORB_GATHER_THREADS"); int v = e ? atoi(e) : kGatherThreads;
                                         return (v >= 192 && v <= 1024 && v % 64 == 0) ? v : kGatherThreads; }();
        const int mode = mode_count ? 2 : ((G.div_is_pow2 && G.fast_norm) ? 1 : 0);
        const float* en = (const float*)c->entries.p;
#define LAUNCH_G(PP, MM, RR) ev_gather_kernel<PP, MM, RR><<<nb, gthreads, 0, c->stream>>>(d_slice_eb, d_order, G, d_tile_cnt, d_tile_base, en, d_f32, d_minmax_enc)
        if (sparse) {
            const uint2* en2 = (const uint2*)c->entries.p;
            // large launches: 8 consecutive work items per wave (a wave per item leaves the launch bound by workgroup dispatch)
            const int per_wave = nb >= 65536 ? 16 : 1;
            const int grid = (nb + kSparseWaves * per_wave - 1) / (kSparseWaves * per_wave);
            if (pol) ev_gather_sparse_kernel<true><<<grid, 64 * kSparseWaves, 0, c->stream>>>(d_slice_eb, G, per_wave, d_tile_cnt, d_tile_base, en2, d_f32, d_minmax_enc);
            else ev_gather_sparse_kernel<false><<<grid, 64 * kSparseWaves, 0, c->stream>>>(d_slice_eb, G, per_wave, d_tile_cnt, d_tile_base, en2, d_f32, d_minmax_enc);
        } else if (raw && mode != 2) {
            // the add wave + four value waves with two tile columns each (shortest chain per batch), or -- when the launch has
            // enough tiles to keep every SIMD busy anyway -- two value waves with four columns each (the rectangle arithmetic is
            // done once per four columns: fewest instructions per batch)
            static const int nc_env = [] { const char* e = getenv("EORB_GATHER_NC"); return e ? atoi(e) : 0; }();
            const int NC = nc_env == 2 || nc_env == 4 ? nc_env : (nb >= 32768 ? 4 : 2);
            const int rthreads = 64 * (1 + 8 / NC);
            const uint2* en2 = (const uint2*)c->entries.p;
            const int ngrid = nb;
#define LAUNCH_R(PP, CC) ev_gather_raw_kernel<PP, CC><<<ngrid, rthreads, 0, c->stream>>>(d_slice_eb, d_order, G, d_tile_cnt, d_tile_base, en2, d_f32, d_minmax_enc)
            if (pol) { if (NC == 4) LAUNCH_R(true, 4); else LAUNCH_R(true, 2); }
            else { if (NC == 4) LAUNCH_R(false, 4); else LAUNCH_R(false, 2); }
#undef LAUNCH_R
        } else if (raw) {                                  // count image of raw events
            if (pol) LAUNCH_G(true, 2, true); else LAUNCH_G(false, 2, true);
        } else if (pol) { if (mode == 2) LAUNCH_G(true, 2, false); else if (mode == 1) LAUNCH_G(true, 1, false); else LAUNCH_G(true, 0, false); }
        else { if (mode == 2) LAUNCH_G(false, 2, false); else if (mode == 1) LAUNCH_G(false, 1, false); else LAUNCH_G(false, 0, false); }
#undef LAUNCH_G
        EORB_LAUNCH_CHECK(c, "ev_gather_kernel");
    }
    if (normalized && d_u8) {
        ProfScope ps(c, "ev_normalize");
        dim3 grid((W * H + 255) / 256 > 64 ? 64 : (W * H + 255) / 256, B);
        ev_normalize_kernel<<<grid, 256, 0, c->stream>>>(d_f32, d_minmax_enc, d_u8, W * H, mode_count);
        EORB_LAUNCH_CHECK(c, "ev_normalize_kernel");
    }
    return EORB_OK;
}

__global__ void ev_mathhash_kernel(int which, uint32_t lo_bits, uint32_t hi_bits, unsigned long long* out)
{
    __shared__ uint64_t tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = kExp2Tab[threadIdx.x];
    __syncthreads();
    unsigned long long h = 0;
    for (uint64_t u = (uint64_t)lo_bits + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; u <= hi_bits;
         u += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t ub = (uint32_t)u;
        const float x = __uint_as_float(ub);
        float y;
        if (which == 0) y = dev_expf_nonpos<true>(-x, tab);
        else if (which == 3) y = dev_tanf(x);
        else if (which == 4) y = dev_atanf(x);
        else { float sn, cs; dev_sincosf(x, &sn, &cs); y = (which == 1) ? sn : cs; }
        h += (((unsigned long long)ub * 0x9E3779B97F4A7C15ull) ^ (unsigned long long)__float_as_uint(y)) * 0xC2B2AE3D27D4EB4Full;
    }
    atomicAdd(out, h);
}

// atan2f over generated (y, x) pairs first .. first + count - 1 (the generator of oracle/orc_math.c: orc_atan2_pair)
__global__ void ev_atan2hash_kernel(unsigned long long first, unsigned long long count, unsigned long long* out)
{
    unsigned long long h = 0;
    for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < count; j += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long i = first + j;
        unsigned long long s = (i + 1) * 0x9E3779B97F4A7C15ull;
        s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
        float y, x;
        if (i & 1) { y = (float)((int)(s & 0xffff) - 32768) / 97.0f; x = (float)((int)((s >> 20) & 0xffff) - 32768) / 89.0f; }
        else { y = __uint_as_float((uint32_t)s); x = __uint_as_float((uint32_t)(s >> 32)); }
        if (y != y || x != x) continue;
        const uint32_t rb = __float_as_uint(dev_atan2f(y, x));
        h += ((i * 0x9E3779B97F4A7C15ull) ^ (unsigned long long)rb) * 0xC2B2AE3D27D4EB4Full;
    }
    atomicAdd(out, h);
}

int ev_mathhash(eorb_ctx* c, int which, uint32_t lo_bits, uint32_t hi_bits, unsigned long long* out)
{
    int rc;
    if ((rc = ensure(c, c->minmax, 64))) return rc;
    EORB_HIP(c, hipMemsetAsync(c->minmax.p, 0, 8, c->stream));
    if (which == 5) ev_atan2hash_kernel<<<2048, 256, 0, c->stream>>>((unsigned long long)lo_bits << 20, ((unsigned long long)(hi_bits - lo_bits) + 1) << 20,
                                                                     (unsigned long long*)c->minmax.p);      // pairs [lo << 20, (hi + 1) << 20)
    else ev_mathhash_kernel<<<2048, 256, 0, c->stream>>>(which, lo_bits, hi_bits, (unsigned long long*)c->minmax.p);
    EORB_LAUNCH_CHECK(c, "ev_mathhash_kernel");
    EORB_HIP(c, hipMemcpyAsync(out, c->minmax.p, 8, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, hipStreamSynchronize(c->stream));
    return EORB_OK;
}

// counts floats in [lo, hi] (positive, by bit pattern) for which the reciprocal/fma quotient differs from IEEE ev / norm
int ev_divcheck(eorb_ctx* c, float lo, float hi, float sigma, unsigned long long* bad_out)
{
    const float sig2 = sigma * sigma;
    const float norm = 2.0f * (float)3.1415926535897932384626433832795 * sig2;
    const float rcp = (float)(1.0 / (double)norm);
    int rc;
    if ((rc = ensure(c, c->minmax, 64))) return rc;
    EORB_HIP(c, hipMemsetAsync(c->minmax.p, 0, 8, c->stream));
    uint32_t lb, hb; memcpy(&lb, &lo, 4); memcpy(&hb, &hi, 4);
    ev_divcheck_kernel<<<2048, 256, 0, c->stream>>>(lb, hb, norm, rcp, (unsigned long long*)c->minmax.p);
    EORB_LAUNCH_CHECK(c, "ev_divcheck_kernel");
    EORB_HIP(c, hipMemcpyAsync(bad_out, c->minmax.p, 8, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, hipStreamSynchronize(c->stream));
    return EORB_OK;
}

int ev_trace_read(unsigned long long* out, int n)
{
#ifdef EORB_TRACE
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * std::min(n, 16 * 64 * 8));
    return 1;
#else
    (void)out; (void)n; return 0;
#endif
}

int ev_diag_read(unsigned long long* out16)
{
#ifdef EORB_DIAG
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_diag), sizeof(unsigned long long) * 16);
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof(z));
    return 1;
#else
    (void)out16; return 0;
#endif
}

int ev_decode_minmax(eorb_ctx* c, const uint32_t* d_enc, float* d_out, int B)
{
    ev_decode_minmax_kernel<<<(2 * B + 63) / 64, 64, 0, c->stream>>>(d_enc, d_out, B);
    EORB_LAUNCH_CHECK(c, "ev_decode_minmax_kernel");
    return EORB_OK;
}

}  // namespace eorb
