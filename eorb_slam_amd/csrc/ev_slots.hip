// ev_slots.hip -- raw sensor events -> Gaussian event image through per-tile SLOT lists (the dense-batch form of ev2im_gauss,
// src/Event/EventConversion.cc:215-269 of the reference; bit-exact w.r.t. its sequential loop).
//
// A raw event is a sensor pixel; its undistorted position, hence its whole (2h+1)^2 stamp (exp_XY2f :59-65), is a function of that
// pixel (MyCalibrator::undistPointMaps, Utils/MyCalibrator.cpp:164-180).  For an 8x8 image tile T only the ~(8+2h)^2 sensor pixels
// whose stamps reach T ever contribute: they are T's SLOTS.  Once per (maps, sigma):
//   slot_tab[sensor pixel]  = { first tile, tiles in x / y, the pixel's slot number in each of its <= 4 tiles }
//   rows[T][slot][64]       = what that sensor pixel adds to each of T's 64 pixels (its stamp value, +0.0f outside the stamp)
// Per batch ("Round 3: a different decomposition"):
//   K0  sl_chunks_kernel       chunk descriptors, the slices' first chunks and entry bases from the slice offsets (a kernel argument)
//   K1a sl_count_lds_kernel    entries of every (chunk, tile): the sensor pixels' tile ranges as a 16-bit table in LDS, one wavefront per
//                              chunk (sl_count_kernel where that table does not fit)
//   K1b sl_scan_kernel         one contiguous event-ordered list per (slice, tile); per-tile weights
//       (RANKS form, the default for sensors of <= 65 535 pixels: the values the counting atomics return are the entries' ranks in the
//                              chunk's runs; they go out with the sensor index as an 8-byte record per event, and the scatter below
//                              -- sl_scatter_pre_kernel -- neither counts nor ranks again)
//   K1c sl_scatter_rank_kernel order-preserving scatter of TWO-BYTE entries (0x1000 | slot number): stable ranks from the return values of the
//                              counting LDS atomics, the chunk tile-sorted in LDS, runs streamed out (sl_scatter_kernel: the ballot form,
//                              kept as the fallback when the device check of the atomics' lane order fails)
//       sl_plan_kernel / sl_tasks_kernel   per tile position its lists sorted longest first, long lists to the buckets of K2h, workgroup
//                              tasks by weight -- on the side stream, beside the scatter
//   K2p sl_gather_kernel       one workgroup per task = tile POSITION: the tile's rows staged in LDS once (<= 64 KB), every wavefront takes
//                              (slice, tile) lists by ticket: lane = pixel, per entry v_readlane (1/4) + v_perm_b32 (row address
//                              { slot, 4 * lane }) + ds_read_b32 + v_add_f32, in list order = event order (newVal = image + val,
//                              :251-254).  x + 0.0f == x bit for bit, so lanes the stamp does not reach keep their value.
//   K2h sl_hot_kernel          long lists, beside K2p on the side stream: the rows in 240 VGPRs, per entry ONE scalar instruction (the
//                              16-bit entry IS the low half of M0 in the VGPR index mode) + one v_add_f32
// Versus the batch pipeline of ev_accum.hip (64-entry batches through value waves and an add wave, ~6.4 wave-instructions and 8
// bytes per entry) an entry costs 2-3 issue slots and 2 bytes; K2p sits on the LDS array (2 cycles per entry and CU), K2h on the
// scalar AND the vector ALU at once (1 cycle per entry and CU each: 78 % / 75 % busy, profiles/r04_*).
#include "eorb_ctx.h"
#include "ev_common.h"
#include "dev_math.h"
#include "sl_hot_asm.h"
#include <algorithm>
#include <memory>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace eorb {

constexpr uint32_t kNoSlot = 0xffu;
// An entry of a (slice, tile) list: 16 bits, kEntryTag | slot.  The tag is what M0[15:12] must hold in the VGPR index mode of
// sl_hot_kernel (SRC0 relative), so that kernel moves an entry to M0 with one scalar instruction; sl_gather_kernel reads the slot byte.
typedef uint16_t slot_entry;
constexpr uint32_t kEntryTag = 0x1000u;
constexpr int kSlotScatWaves = 8;
constexpr uint16_t kNoGeo = 0xffffu;           // (tile 127,127 with a second tile: never a valid range, TX and TY <= 127 there)
constexpr int kCountWaves = 16;

// ---- tables ---------------------------------------------------------------------------------------------------------------------
// Tile range of a sensor pixel = tiles that hold an IN-IMAGE pixel of its stamp (:250 `if (isInImage)`): every entry of a list
// then visits at least one pixel, so "the tile was visited" (resolveMinMaxVals :32-39) is "its list is not empty".
__global__ void sl_assign_kernel(const uint32_t* __restrict__ src_info, int nsrc, int W, int H, int h, int TX,
                                 uint32_t* __restrict__ tile_nslots, uint2* __restrict__ slot_tab, uint16_t* __restrict__ slot_geo,
                                 int* __restrict__ info)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsrc) { if (i == nsrc) slot_geo[i] = kNoGeo; return; }      // (pad: the table is copied to LDS as dwords)
    const uint32_t w = src_info[i];
    const int xi = (int)(int16_t)(w & 0xffff), yi = (int)(int16_t)(w >> 16);
    uint2 o = make_uint2(0u, 0xffffffffu);
    if (xi != -32768) {
        const int x0 = max(xi - h, 0), x1 = min(xi + h, W - 1), y0 = max(yi - h, 0), y1 = min(yi + h, H - 1);
        if (x1 >= x0 && y1 >= y0) {
            const int tx0 = x0 >> 3, tx1 = x1 >> 3, ty0 = y0 >> 3, ty1 = y1 >> 3;
            const int nx = tx1 - tx0 + 1, ny = ty1 - ty0 + 1;           // 1 or 2 (h <= 4)
            uint32_t slots = 0xffffffffu;
            for (int dy = 0; dy < ny; dy++)
                for (int dx = 0; dx < nx; dx++) {
                    uint32_t s = atomicAdd(&tile_nslots[(ty0 + dy) * TX + tx0 + dx], 1u);
                    if (s >= kNoSlot) { atomicOr(&info[2], 1); s = 0; }
                    const int sh = 8 * (dy * 2 + dx);
                    slots = (slots & ~(0xffu << sh)) | (s << sh);
                }
            o.x = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)nx << 16) | ((uint32_t)ny << 18);
            o.y = slots;
        }
    }
    slot_tab[i] = o;
    // the tile range alone in 16 bits, for the count pass (its table lives in LDS): first tile x | y << 7 | two tiles in x << 14 | in y << 15
    slot_geo[i] = o.x ? (uint16_t)((o.x & 0x7f) | (((o.x >> 8) & 0x7f) << 7) | ((((o.x >> 16) & 3) - 1) << 14) | ((((o.x >> 18) & 3) - 1) << 15)) : kNoGeo;
}

// rowbase[t] = first row of tile t in the row table; info[0] = rows in all, info[1] = most slots of a tile
__global__ __launch_bounds__(1024) void sl_rowbase_kernel(const uint32_t* __restrict__ nslots, int NT, uint32_t* __restrict__ rowbase,
                                                          int* __restrict__ info)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t wmax[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (NT + 1023) / 1024;
    const int t0 = tid * per, t1 = min(t0 + per, NT);
    uint32_t mine = 0, mx = 0;
    for (int t = t0; t < t1; t++) { const uint32_t v = nslots[t]; mine += v; mx = max(mx, v); }
    uint32_t incl = (uint32_t)wave_incl_scan((int)mine);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (lane == 63) wsum[wave] = incl;
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    uint32_t before = incl - mine;
    for (int w = 0; w < wave; w++) before += wsum[w];
    for (int t = t0; t < t1; t++) { rowbase[t] = before; before += nslots[t]; }
    if (tid == 1023) {
        uint32_t m = 0;
        for (int w = 0; w < 16; w++) m = max(m, wmax[w]);
        info[0] = (int)before; info[1] = (int)m;
    }
}

// one wavefront per sensor pixel: lane = pixel of the tile; rows[(rowbase[tile] + slot) * 64 + lane] = the stamp tap that falls on
// that pixel (stamps[src][x offset][y offset], ev_stamp_kernel), +0.0f outside the stamp or the image
__global__ __launch_bounds__(256) void sl_rows_kernel(const uint32_t* __restrict__ src_info, const uint2* __restrict__ slot_tab, int nsrc,
                                                      int W, int H, int h, int TX, const float* __restrict__ stamps, int stamp_stride,
                                                      int SWP, const float2* __restrict__ lut, float two_sig2, float norm, const uint32_t* __restrict__ rowbase, float* __restrict__ rows)
{
    __shared__ uint64_t tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = kExp2Tab[threadIdx.x];
    __syncthreads();
    const int src = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (src >= nsrc) return;
    const uint2 st = slot_tab[src];
    const int nx = (st.x >> 16) & 3, ny = (st.x >> 18) & 3;
    if (!nx || !ny) return;
    const uint32_t w = src_info[src];
    const int xi = (int)(int16_t)(w & 0xffff), yi = (int)(int16_t)(w >> 16);
    const int tx0 = st.x & 0xff, ty0 = (st.x >> 8) & 0xff;
    for (int dy = 0; dy < ny; dy++)
        for (int dx = 0; dx < nx; dx++) {
            const uint32_t slot = (st.y >> (8 * (dy * 2 + dx))) & 0xffu;
            const int tile = (ty0 + dy) * TX + tx0 + dx;
            const int px = (tx0 + dx) * 8 + (lane & 7), py = (ty0 + dy) * 8 + (lane >> 3);
            const int i = px - xi + h, j = py - yi + h;
            float v = 0.0f;
            if (i >= 0 && i <= 2 * h && j >= 0 && j <= 2 * h && px < W && py < H) {
                if (stamps) v = stamps[(size_t)src * stamp_stride + i * SWP + j];
                else {
                    // no stamp table (the per-call positions of float events): the tap as ev_stamp_kernel evaluates it, exp_XY2f :59-65
                    const float2 q = lut[src];
                    const float xr = q.x - (float)xi, yr = q.y - (float)yi;
                    const float fx = (float)(i - h) - xr, fy = (float)(j - h) - yr;
                    const float xx = fx * fx, yy = fy * fy;
                    float dd = xx + yy;
                    dd = dd / two_sig2;
                    v = dev_expf_nonpos<true>(-dd, tab) / norm;
                }
            }
            rows[((size_t)rowbase[tile] + slot) * 64 + lane] = v;
        }
}

// ---- K0: the chunk descriptors, the slices' first chunks and entry bases, from the slice offsets (a kernel argument: no staging
// buffer, no host-to-device copy in front of the count pass).  Same arithmetic as the host loop of ev_slots_accumulate. ----
constexpr int kOffsetsInArg = 256;
struct SliceOffsets { int64_t off[kOffsetsInArg + 1]; };
__global__ __launch_bounds__(256) void sl_chunks_kernel(SliceOffsets S, int B, int chunk_shift /* chunks are 2^shift events */, int NT, int nchunks,
                                                        ChunkDesc* __restrict__ chunks, int* __restrict__ slice_c0, int64_t* __restrict__ slice_eb)
{
    __shared__ int c0[kOffsetsInArg + 1];
    __shared__ long long e0[kOffsetsInArg + 1];
    const int tid = threadIdx.x;
    const int64_t chunk = (int64_t)1 << chunk_shift;
    // thread b = slice b: S is the FIRST kernel argument, read per lane straight from the kernarg segment (a loop over S.off[b] is a
    // chain of 2 B dependent scalar loads: 14 us for 128 slices)
    const int64_t* off = (const int64_t*)__builtin_amdgcn_kernarg_segment_ptr();
    const int64_t n = tid < B ? off[tid + 1] - off[tid] : 0;
    c0[tid + 1] = tid < B ? (int)((n + chunk - 1) >> chunk_shift) : 0;
    e0[tid + 1] = tid < B ? ((n * 4 + (int64_t)NT * 16 + 15) & ~(int64_t)15) : 0;     // (bases are multiples of 16: aligning the sum = summing the aligned sizes)
    if (tid == 0) { c0[0] = 0; e0[0] = 0; }
    __syncthreads();
    for (int d = 1; d < kOffsetsInArg; d <<= 1) {                                     // inclusive scans over [1 .. 256]
        const int a = tid + 1 > d ? c0[tid + 1 - d] : 0;
        const long long b = tid + 1 > d ? e0[tid + 1 - d] : 0;
        __syncthreads();
        c0[tid + 1] += a; e0[tid + 1] += b;
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        if (tid < B) slice_eb[tid] = e0[tid];
        for (int b = tid; b <= B; b += blockDim.x) slice_c0[b] = c0[b];
    }
    const int g = blockIdx.x * blockDim.x + tid;
    if (g >= nchunks) return;
    int lo = 0, hi = B;                      // the slice whose chunks hold g: c0[lo] <= g < c0[lo + 1] (empty slices own no chunk)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c0[mid] <= g) lo = mid; else hi = mid; }
    const int64_t start = off[lo] + ((int64_t)(g - c0[lo]) << chunk_shift);
    ChunkDesc cd; cd.start = start; cd.n = (int32_t)min(chunk, off[lo + 1] - start); cd.slice = lo;
    chunks[g] = cd;
}

// ---- K1a: entries of every (chunk, tile) -----------------------------------------------------------------------------------------
// (stride: 16 = eorb_raw_event, 4 = eorb_raw_event4, 2 = eorb_raw_event2, -4 = hashed records; a template parameter in the two hot kernels)
// the word of record k that names the sensor pixel: the first dword of the 16- / 4-byte records, the 16-bit linear index of the 2-byte one
template <int stride>
__device__ __forceinline__ uint32_t sl_load_rec(const unsigned char* e, int k)
{
    if (stride == 2) { const uint32_t v = *(const uint16_t*)(e + (size_t)k * 2); return v == 0xffffu ? 0x7fffffffu : v; }      // (0xffff: no event)
    return *(const uint32_t*)(e + (size_t)k * (size_t)(stride < 0 ? -stride : stride));
}
template <int stride>
__global__ __launch_bounds__(256) void sl_count_kernel(const eorb_raw_event* __restrict__ ev, const ChunkDesc* __restrict__ chunks,
                                                       const uint2* __restrict__ slot_tab, int LW, int LH, int TX, int NT,
                                                       uint16_t* __restrict__ segcnt)
{
    extern __shared__ uint32_t cnt[];               // NT
    const ChunkDesc cd = chunks[blockIdx.x];
    for (int i = threadIdx.x; i < NT; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const unsigned char* e = (const unsigned char*)ev + (size_t)cd.start * (size_t)(stride < 0 ? -stride : stride);      // 16-byte eorb_raw_event, 4-byte eorb_raw_event4, 4-byte hashed, 2-byte eorb_raw_event2
    const uint32_t xmask = stride == 4 ? 0x7fffu : 0xffffu;
    // records that name a table row instead of (x, y): stride -4 = { row of the position's table entry | polarity << 31 } (float events in
    // bulk), stride 2 = the sensor pixel's linear index y * LW + x (the 2-byte wire record)
    const bool hashed = stride < 0 || stride == 2;
    constexpr int U = 8;
    for (int k0 = threadIdx.x; k0 < cd.n; k0 += blockDim.x * U) {
        uint32_t xy[U], rg[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int k = k0 + u * blockDim.x; xy[u] = k < cd.n ? sl_load_rec<stride>(e, k) : 0xffffffffu; }
        // (whole 16-byte records per lane, non-temporal, were measured: 0.69 ms instead of 0.62)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int x = (int)(xy[u] & xmask), y = (int)(xy[u] >> 16);
            const uint32_t row = xy[u] & 0x7fffffffu;                    // hashed records: the row of the event's position
            if (hashed) rg[u] = (row != 0x7fffffffu && row < (uint32_t)LW * (uint32_t)LH) ? slot_tab[row].x : 0u;
            else rg[u] = (x < LW && y < LH) ? slot_tab[(uint32_t)y * (uint32_t)LW + x].x : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int nx = (rg[u] >> 16) & 3, ny = (rg[u] >> 18) & 3;
            const int t0 = (int)((rg[u] >> 8) & 0xff) * TX + (int)(rg[u] & 0xff);
            if (nx && ny) {
                atomicAdd(&cnt[t0], 1u);
                if (nx > 1) atomicAdd(&cnt[t0 + 1], 1u);
                if (ny > 1) { atomicAdd(&cnt[t0 + TX], 1u); if (nx > 1) atomicAdd(&cnt[t0 + TX + 1], 1u); }
            }
        }
    }
    __syncthreads();
    uint16_t* sc = segcnt + (size_t)blockIdx.x * NT;
    for (int i = threadIdx.x; i < NT; i += blockDim.x) sc[i] = (uint16_t)cnt[i];
}

// K1a with the tile ranges in LDS.  The count pass is a streaming read of the events plus ONE table lookup per event; with the table
// in global memory every lookup of a random sensor pixel is its own L2 request (345 KB of uint2: no L1 hit rate to speak of) and the
// pass ran at 3.2 TB/s, with the lookup computed instead it ran at 5.2 (profiles/r03_ko_table_lookup.txt).  So: one 16-wave
// workgroup per CU copies the 16-bit ranges of all sensor pixels into LDS once (240x180: 86 KB) and every WAVEFRONT counts whole
// chunks on its own -- own counters (packed 16-bit, a chunk has <= 2048 events), no workgroup barrier after the table is in.
// KEYED (stride 16 only): the records are float events (eorb_event16) whose positions are looked up in the context's frozen position
// dictionary -- a compact open-addressing table { x bits, y bits, dense id, - } -- and the dense id goes out as a 4-byte hashed record
// for the scatter (rec[event index]); a position the dictionary does not hold counts a miss (the caller then tabulates the call's
// positions afresh).  One pass over the events instead of a hashing pass plus a counting pass.
__device__ __forceinline__ uint32_t sl_dict_hash(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}
// rec16 (stride 16 / 4 on sensors of at most 65 535 pixels): the pass also writes every event's linear sensor index y * LW + x as a
// 2-byte record (0xffff: none), which is all the scatter needs: it then reads 2 bytes per event instead of a dword out of every 16-byte
// record (2.05 GB of the step's HBM traffic for 0.26 + 0.26).
// RANKS: the pass keeps what its counting atomics return.  One wavefront counts a chunk, its LDS instructions execute in order and the
// lanes of one ds_add_rtn that hit the same counter are served in lane order (the property sl_rankcheck_kernel verifies, see
// sl_scatter_rank_kernel) -- so, with an event's tiles visited by parity class like there, the returned value IS the entry's rank in
// the chunk's (tile) run.  It goes out with the sensor index as one 8-byte record { index : 16, rank of the event's tile of class j :
// 12 bits each }: the scatter (sl_scatter_pre_kernel) then neither counts nor ranks again -- its two most expensive phases.
template <int stride, bool KEYED = false, bool RANKS = false>
__global__ __launch_bounds__(64 * kCountWaves) void sl_count_lds_kernel(const eorb_raw_event* __restrict__ ev, const ChunkDesc* __restrict__ chunks,
                                                                       int nchunks, const uint16_t* __restrict__ slot_geo, int LW, int LH,
                                                                       int TX, int NT, uint16_t* __restrict__ segcnt, SlotDict D = SlotDict{nullptr, 0u, nullptr, nullptr, 0},
                                                                       uint16_t* __restrict__ rec16 = nullptr)
{
    extern __shared__ uint32_t smc[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntab = (LW * LH + 2) >> 1, NTp = (NT + 1) & ~1;
    for (int i = tid; i < ntab; i += 64 * kCountWaves) smc[i] = ((const uint32_t*)slot_geo)[i];
    const uint16_t* tab = (const uint16_t*)smc;
    uint32_t* cnt = smc + ntab + wave * (NTp >> 1);
    __syncthreads();
    constexpr int astride = stride < 0 ? -stride : stride;
    const uint32_t xmask = stride == 4 ? 0x7fffu : 0xffffu;
    constexpr int U = 8;
    for (int ch = blockIdx.x * kCountWaves + wave; ch < nchunks; ch += gridDim.x * kCountWaves) {
        const ChunkDesc cd = chunks[ch];
        for (int i = lane; i < (NTp >> 1); i += 64) cnt[i] = 0u;
        const unsigned char* e = (const unsigned char*)ev + (size_t)cd.start * (size_t)astride;
        uint32_t xyn[U];                                               // (the next trip's records, requested before this trip's are counted)
        if (!KEYED) {
#pragma unroll
            for (int u = 0; u < U; u++) { const int k = lane + u * 64; xyn[u] = k < cd.n ? sl_load_rec<stride>(e, k) : 0xffffffffu; }
        }
        for (int k0 = lane; k0 < cd.n; k0 += 64 * U) {
            uint32_t xy[U], g[U];
            uint32_t r16[U];                                           // (RANKS: the events' sensor indices)
            if (KEYED) {
                // (the first probe's entry as three scalars per event: with a uint4 ent[U] copied into the probe loop's variable the compiler
                // merged that copy and the loop's load into ONE load through a pointer phi -- the array went to scratch, 16 bytes stored and
                // read back through a flat load per event: 2.04 ms for this pass)
                uint2 pos[U]; uint32_t ex[U], ey[U], ez[U], hh[U];
#pragma unroll
                for (int u = 0; u < U; u++) { const int k = k0 + u * 64; pos[u] = k < cd.n ? *(const uint2*)(e + (size_t)k * 16) : make_uint2(0x7fc00000u, 0x7fc00000u); }
#pragma unroll
                for (int u = 0; u < U; u++) { hh[u] = sl_dict_hash((uint64_t)pos[u].x | ((uint64_t)pos[u].y << 32)) & D.mask; const uint4 t = D.hash[hh[u]]; ex[u] = t.x; ey[u] = t.y; ez[u] = t.z; }      // (a non-temporal load here: 1.89 -> 3.09 ms; the L1 does serve part of the probes)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int k = k0 + u * 64;
                    const float fx = __uint_as_float(pos[u].x), fy = __uint_as_float(pos[u].y);
                    uint32_t id = 0x7fffffffu;                           // dropped (NaN coordinates are never in the image)
                    if (k < cd.n && fx == fx && fy == fy) {
                        uint32_t h = hh[u], qx = ex[u], qy = ey[u], qz = ez[u]; int probes = 0;
                        while (!(qx == pos[u].x && qy == pos[u].y) && qz != 0xffffffffu && ++probes <= 64) { h = (h + 1) & D.mask; const uint4 t = D.hash[h]; qx = t.x; qy = t.y; qz = t.z; }
                        if (qx == pos[u].x && qy == pos[u].y && qz != 0xffffffffu) id = qz;
                        else atomicAdd(D.miss, 1);
                    }
                    if (RANKS) r16[u] = id < (uint32_t)LW * (uint32_t)LH ? id : 0xffffu;      // (the ranked record below carries the id)
                    else if (k < cd.n) {
                        if (D.rec2) ((uint16_t*)D.rec)[cd.start + k] = id == 0x7fffffffu ? (uint16_t)0xffffu : (uint16_t)id;
                        else D.rec[cd.start + k] = id;
                    }
                    g[u] = id < (uint32_t)LW * (uint32_t)LH ? tab[id] : kNoGeo;
                }
            } else {
#pragma unroll
            for (int u = 0; u < U; u++) xy[u] = xyn[u];
            if (k0 + 64 * U < cd.n) {
#pragma unroll
                for (int u = 0; u < U; u++) { const int k = k0 + 64 * U + u * 64; xyn[u] = k < cd.n ? sl_load_rec<stride>(e, k) : 0xffffffffu; }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t x = xy[u] & xmask, y = xy[u] >> 16, row = xy[u] & 0x7fffffffu;      // (hashed records: the row itself)
                if (stride < 0 || stride == 2) g[u] = row < (uint32_t)LW * (uint32_t)LH ? tab[row] : kNoGeo;
                else g[u] = (x < (uint32_t)LW && y < (uint32_t)LH) ? tab[y * (uint32_t)LW + x] : kNoGeo;
                if (RANKS) {
                    if (stride < 0 || stride == 2) r16[u] = row < (uint32_t)LW * (uint32_t)LH ? row : 0xffffu;
                    else r16[u] = (x < (uint32_t)LW && y < (uint32_t)LH) ? y * (uint32_t)LW + x : 0xffffu;
                } else if ((stride == 16 || stride == 4) && rec16) {
                    const int k = k0 + u * 64;
                    if (k < cd.n) rec16[cd.start + k] = (x < (uint32_t)LW && y < (uint32_t)LH) ? (uint16_t)(y * (uint32_t)LW + x) : (uint16_t)0xffffu;
                }
            }
            }
            if (RANKS) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    uint64_t rk = 0ull;
                    if (g[u] != kNoGeo) {
                        const int tx0 = (int)(g[u] & 0x7f), ty0 = (int)((g[u] >> 7) & 0x7f);
                        const int nx = 1 + (int)((g[u] >> 14) & 1), ny = 1 + (int)((g[u] >> 15) & 1);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int dx = ((j & 1) - tx0) & 1, dy = ((j >> 1) - ty0) & 1;       // the event's tile of class j, if it has one
                            if (dx < nx && dy < ny) {
                                const int t = (ty0 + dy) * TX + tx0 + dx;
                                const uint32_t o = atomicAdd(&cnt[t >> 1], 1u << (16 * (t & 1)));
                                rk |= (uint64_t)((o >> (16 * (t & 1))) & 0xfffu) << (12 * j);
                            }
                        }
                    }
                    const int k = k0 + u * 64;
                    if (k < cd.n) ((uint64_t*)rec16)[cd.start + k] = (uint64_t)(r16[u] & 0xffffu) | (rk << 16);
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (g[u] != kNoGeo) {
                    const int t0 = (int)((g[u] >> 7) & 0x7f) * TX + (int)(g[u] & 0x7f);
                    const bool two_x = (g[u] >> 14) & 1, two_y = (g[u] >> 15) & 1;
                    atomicAdd(&cnt[t0 >> 1], 1u << (16 * (t0 & 1)));
                    if (two_x) atomicAdd(&cnt[(t0 + 1) >> 1], 1u << (16 * ((t0 + 1) & 1)));
                    if (two_y) {
                        atomicAdd(&cnt[(t0 + TX) >> 1], 1u << (16 * ((t0 + TX) & 1)));
                        if (two_x) atomicAdd(&cnt[(t0 + TX + 1) >> 1], 1u << (16 * ((t0 + TX + 1) & 1)));
                    }
                }
            }
        }
        // (one wavefront: its LDS instructions execute in order, nothing to wait for but the compiler must keep that order)
        __builtin_amdgcn_wave_barrier();
        uint16_t* sc = segcnt + (size_t)ch * NT;
        const uint16_t* c16 = (const uint16_t*)cnt;
        for (int i = lane; i < NT; i += 64) sc[i] = c16[i];
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- K1b: one workgroup per slice.  Per tile: exclusive scan of its counts over the slice's chunks (segbase), the total (tile_cnt);
// exclusive scan over the tiles of the totals rounded up to 16 entries (tile_base: every list starts on a 16-byte boundary) ----
__global__ __launch_bounds__(1024) void sl_scan_kernel(const int* __restrict__ slice_chunk0, const uint16_t* __restrict__ segcnt, int NT,
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
        const uint32_t padded = (run + 15u) & ~15u;
        uint32_t incl = (uint32_t)wave_incl_scan((int)padded);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (tile < NT) tile_base[(size_t)slice * NT + tile] = before + incl - padded;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry = before + incl;
        __syncthreads();
    }
}

// ---- K1c: order-preserving scatter of 2-byte entries.  One 8-wave workgroup per chunk (<= 2048 events); wave w owns the w-th
// share of the chunk's events.
//   A  slot bytes / first tile of every event into LDS, tile ranges into registers, counts per (wave, tile) by LDS atomics
//   B  per tile the exclusive prefix over the waves; exclusive scan of the totals over the tiles: loff[t] = start of tile t's run
//   C  every wave walks its sub-batches of 64 events in order: rank among the lanes of the same tile by ballot matching (tiles
//      visited in parity classes: a lane has at most one tile per class) + the wave's running counter: sidx[slot of the sorted
//      order] = event | dx << 11 | dy << 13
//   D  slot p -> (event, tile) -> entry = the event's slot number in that tile << 8, stored at the run's place in the tile's
//      global list: consecutive threads write consecutive entries of a run ----
__global__ __launch_bounds__(64 * kSlotScatWaves) void sl_scatter_kernel(const eorb_raw_event* __restrict__ ev, const ChunkDesc* __restrict__ chunks,
                                                                         const uint2* __restrict__ slot_tab, int stride, int LW, int LH, int TX, int TY,
                                                                         int NT, int chunk_cap, const int64_t* __restrict__ slice_ebase,
                                                                         const uint32_t* __restrict__ segbase, const uint32_t* __restrict__ tile_base,
                                                                         slot_entry* __restrict__ entries)
{
    extern __shared__ unsigned char sm2[];
    __shared__ uint32_t s_wsum[kSlotScatWaves];
    constexpr int NTHR = 64 * kSlotScatWaves;
    constexpr int R = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x;
    const ChunkDesc cd = chunks[chunk];
    const int NTp = (NT + 1) & ~1;
    uint32_t* pay = (uint32_t*)sm2;                                   // chunk_cap: the event's slot bytes
    uint16_t* prng = (uint16_t*)(pay + chunk_cap);                    // chunk_cap: first tile of the event's range, tx0 | ty0 << 8
    uint16_t* sidx = prng + chunk_cap;                                // chunk_cap * 4
    uint16_t* cntw = sidx + (size_t)chunk_cap * R * R;                // kSlotScatWaves * NTp
    uint16_t* loff = cntw + kSlotScatWaves * NTp;                     // NT + 1 (+ 1 pad)
    uint32_t* gbase = (uint32_t*)(loff + NTp + 2);                    // NT
    for (int i = tid; i < kSlotScatWaves * NTp / 2; i += NTHR) ((uint32_t*)cntw)[i] = 0u;
    const unsigned char* e = (const unsigned char*)ev + (size_t)cd.start * (size_t)(stride < 0 ? -stride : stride);      // 16-byte eorb_raw_event, 4-byte eorb_raw_event4, 4-byte hashed, 2-byte eorb_raw_event2
    const uint32_t xmask = stride == 4 ? 0x7fffu : 0xffffu;
    const bool hashed = stride < 0 || stride == 2;          // records that name a table row (see sl_count_kernel)
    const int Q = (((cd.n + kSlotScatWaves - 1) / kSlotScatWaves) + 63) & ~63;
    const int S = Q >> 6;
    constexpr int SMAX = 4;
    uint32_t rng[SMAX];                                               // slot_tab.x of the event; 0 = no entry
    __syncthreads();
    // ---- A ----
    uint32_t* cw32 = (uint32_t*)(cntw + wave * NTp);
    uint32_t rsrc[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        const int k = wave * Q + s * 64 + lane;
        uint32_t q = 0xffffffffu;
        if (s < S && k < cd.n) q = stride == 2 ? sl_load_rec<2>(e, k) : *(const uint32_t*)(e + (size_t)k * (size_t)(stride < 0 ? -stride : stride));
        const int x = q == 0xffffffffu ? 0xffff : (int)(q & xmask), y = (int)(q >> 16);
        if (hashed) { const uint32_t row = q & 0x7fffffffu; rsrc[s] = (row != 0x7fffffffu && row < (uint32_t)LW * (uint32_t)LH) ? row : 0xffffffffu; }
        else rsrc[s] = (x < LW && y < LH) ? (uint32_t)y * (uint32_t)LW + x : 0xffffffffu;
    }
    uint2 rst[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; s++) rst[s] = rsrc[s] != 0xffffffffu ? slot_tab[rsrc[s]] : make_uint2(0u, 0xffffffffu);
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        rng[s] = 0u;
        const int k = wave * Q + s * 64 + lane;
        if (s < S && k < cd.n) {
            const uint32_t rg = rst[s].x;
            const int nx = (rg >> 16) & 3, ny = (rg >> 18) & 3;
            pay[k] = rst[s].y;
            prng[k] = (uint16_t)(rg & 0xffff);
            if (nx && ny) {
                rng[s] = rg;
                const int t0 = (int)((rg >> 8) & 0xff) * TX + (int)(rg & 0xff);
                for (int dy = 0; dy < ny; dy++)
                    for (int dx = 0; dx < nx; dx++) {
                        const int t = t0 + dy * TX + dx;
                        atomicAdd(&cw32[t >> 1], 1u << (16 * (t & 1)));       // 16-bit counters, two per word (a share has <= 256 events)
                    }
            }
        }
    }
    __syncthreads();
    // ---- B ----
    {
        const int per = (NT + NTHR - 1) / NTHR;
        const int t0 = tid * per, t1 = min(t0 + per, NT);
        uint32_t mine = 0;
        for (int t = t0; t < t1; t++) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < kSlotScatWaves; w++) { const uint32_t v = cntw[w * NTp + t]; cntw[w * NTp + t] = (uint16_t)run; run += v; }
            loff[t] = (uint16_t)run;
            mine += run;
            gbase[t] = tile_base[(size_t)cd.slice * NT + t] + segbase[(size_t)chunk * NT + t];
        }
        uint32_t incl = (uint32_t)wave_incl_scan((int)mine);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w = 0; w < wave; w++) before += s_wsum[w];
        for (int t = t0; t < t1; t++) { const uint32_t v = loff[t]; loff[t] = (uint16_t)before; before += v; }
        if (tid == NTHR - 1) loff[NT] = (uint16_t)before;
    }
    __syncthreads();
    // ---- C ----
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint16_t* cw = cntw + wave * NTp;
    const int txr = (TX + R - 1) / R;
    int mbits = 1; while ((1 << mbits) < txr * ((TY + R - 1) / R)) mbits++;
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        if (s >= S) break;
        const uint32_t rg = rng[s];
        const int tx0 = rg & 0xff, ty0 = (rg >> 8) & 0xff, tx1 = tx0 + (int)((rg >> 16) & 3) - 1, ty1 = ty0 + (int)((rg >> 18) & 3) - 1;
        const bool valid = rg != 0u;
        const uint16_t kloc = (uint16_t)(wave * Q + s * 64 + lane);
#pragma unroll
        for (int cy = 0; cy < R; cy++) {
#pragma unroll
            for (int cx = 0; cx < R; cx++) {
                const int tx = tx0 + ((cx - tx0 % R) + R) % R;
                const int ty = ty0 + ((cy - ty0 % R) + R) % R;
                const bool has = valid && tx <= tx1 && ty <= ty1;
                uint64_t m = __ballot(has);
                if (m == 0ull) continue;
                const int key = has ? ty * TX + tx : 0;
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
    // ---- D ----
    slot_entry* out = entries + (size_t)slice_ebase[cd.slice];
    const int E = loff[NT];
    for (int p = tid; p < E; p += NTHR) {
        const uint32_t sv = sidx[p];
        const int k = sv & 0x7ff;
        const int dx = (sv >> 11) & 3, dy = (sv >> 13) & 3;
        const uint32_t r0 = prng[k];
        const int t = ((int)(r0 >> 8) + dy) * TX + (int)(r0 & 0xff) + dx;
        const uint32_t slot = (pay[k] >> (8 * (dy * 2 + dx))) & 0xffu;
        out[(size_t)gbase[t] + (uint32_t)(p - (int)loff[t])] = (slot_entry)(kEntryTag | slot);
    }
}

// ---- K1c, rank form: the stable rank of an entry inside its wavefront's share of a tile comes back from the LDS atomic that counts
// it.  Lanes of one ds_add_rtn that hit the same word are served in lane order (not an architectural promise: sl_rankcheck_kernel
// verifies it on the device once per context, and the ballot form above stays as the fallback), instructions of a wave execute
// in order, and the waves' shares are ordered by the prefix over the waves -- so no ballots, no parity classes, no per-event list
// of sorted slots: phase A keeps four 8-bit ranks per event, phase C writes the entry byte and its tile straight to their place
// in the chunk's tile-sorted order, phase D streams that order out run by run.
template <int stride, int NW /* wavefronts = 256-event shares of a chunk: 8 (2 048 events) or 16 (4 096) */>
__global__ __launch_bounds__(64 * NW) void sl_scatter_rank_kernel(const eorb_raw_event* __restrict__ ev, const ChunkDesc* __restrict__ chunks,
                                                                              const uint2* __restrict__ slot_tab, int LW, int LH, int TX, int NT,
                                                                              int chunk_cap, const int64_t* __restrict__ slice_ebase,
                                                                              const uint32_t* __restrict__ segbase, const uint32_t* __restrict__ tile_base,
                                                                              slot_entry* __restrict__ entries)
{
    extern __shared__ unsigned char sm2[];
    __shared__ uint32_t s_wsum[NW];
    constexpr int NTHR = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x;
    const ChunkDesc cd = chunks[chunk];
    const int NTp = (NT + 1) & ~1;
    uint32_t* gbase = (uint32_t*)sm2;                                 // NT: first entry of the tile's run in the global lists, minus loff
    uint16_t* stile = (uint16_t*)(gbase + NT);                        // chunk_cap * 4: tile of every slot of the sorted order
    uint16_t* cntw = stile + (size_t)chunk_cap * 4;                   // NW * NTp
    uint16_t* loff = cntw + NW * NTp;                     // NT + 1 (+ 1 pad)
    uint8_t* sorted = (uint8_t*)(loff + NTp + 2);                     // chunk_cap * 4: the entry bytes in tile-sorted order
    for (int i = tid; i < NW * NTp / 2; i += NTHR) ((uint32_t*)cntw)[i] = 0u;
    const unsigned char* e = (const unsigned char*)ev + (size_t)cd.start * (size_t)(stride < 0 ? -stride : stride);      // 16-byte eorb_raw_event, 4-byte eorb_raw_event4, 4-byte hashed, 2-byte eorb_raw_event2
    const uint32_t xmask = stride == 4 ? 0x7fffu : 0xffffu;
    const bool hashed = stride < 0 || stride == 2;          // records that name a table row (see sl_count_kernel)
    const int Q = (((cd.n + NW - 1) / NW) + 63) & ~63;
    const int S = Q >> 6;
    constexpr int SMAX = 4;
    __syncthreads();
    // ---- A: tile ranges and slot bytes into registers; every entry counted, its rank (< 256: a share has <= 256 events) kept ----
    uint32_t* cw32 = (uint32_t*)(cntw + wave * NTp);
    uint32_t rsrc[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        const int k = wave * Q + s * 64 + lane;
        const uint32_t q = (s < S && k < cd.n) ? sl_load_rec<stride>(e, k) : 0xffffffffu;
        const int x = q == 0xffffffffu ? 0xffff : (int)(q & xmask), y = (int)(q >> 16);
        if (hashed) { const uint32_t row = q & 0x7fffffffu; rsrc[s] = (row != 0x7fffffffu && row < (uint32_t)LW * (uint32_t)LH) ? row : 0xffffffffu; }
        else rsrc[s] = (x < LW && y < LH) ? (uint32_t)y * (uint32_t)LW + x : 0xffffffffu;
    }
    uint2 rst[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; s++) rst[s] = rsrc[s] != 0xffffffffu ? slot_tab[rsrc[s]] : make_uint2(0u, 0xffffffffu);
    // Tiles are visited in parity classes (tx & 1, ty & 1): a tile belongs to one class, so all lanes of a sub-batch that target it do
    // so in the SAME atomic instruction (an event's second tile must not be counted after a later event's first one).
    uint32_t rk[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        const uint32_t rg = rst[s].x;
        const int nx = (rg >> 16) & 3, ny = (rg >> 18) & 3;
        const int tx0 = rg & 0xff, ty0 = (rg >> 8) & 0xff;
        rk[s] = 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
#ifdef EORB_TEST_RANK_ORDER_BUG        // (test builds only: the naive order, to show that the parity tests catch it)
            const int dx = j & 1, dy = j >> 1;
#else
            const int dx = ((j & 1) - tx0) & 1, dy = ((j >> 1) - ty0) & 1;       // the event's tile of class j, if it has one
#endif
            if (nx && ny && dx < nx && dy < ny) {
                const int t = (ty0 + dy) * TX + tx0 + dx;
                const uint32_t o = atomicAdd(&cw32[t >> 1], 1u << (16 * (t & 1)));
                rk[s] |= ((o >> (16 * (t & 1))) & 0xffu) << (8 * j);
            }
        }
    }
    __syncthreads();
#if defined(EORB_KO_SCAT) && EORB_KO_SCAT == 1          // (experiment builds: the kernel cut short after a phase)
    if (rk[0] == 0x12345678u) entries[0] = 0; return;
#endif
    // ---- B: per tile the exclusive prefix over the waves; exclusive scan of the totals over the tiles ----
    {
        const int per = (NT + NTHR - 1) / NTHR;
        const int t0 = tid * per, t1 = min(t0 + per, NT);
        uint32_t mine = 0;
        for (int t = t0; t < t1; t++) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) { const uint32_t v = cntw[w * NTp + t]; cntw[w * NTp + t] = (uint16_t)run; run += v; }
            loff[t] = (uint16_t)run;
            mine += run;
        }
        uint32_t incl = (uint32_t)wave_incl_scan((int)mine);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w = 0; w < wave; w++) before += s_wsum[w];
        for (int t = t0; t < t1; t++) {
            const uint32_t v = loff[t]; loff[t] = (uint16_t)before;
            gbase[t] = tile_base[(size_t)cd.slice * NT + t] + segbase[(size_t)chunk * NT + t] - before;
            before += v;
        }
        if (tid == NTHR - 1) loff[NT] = (uint16_t)before;
    }
    __syncthreads();
#if defined(EORB_KO_SCAT) && EORB_KO_SCAT == 2
    if (rk[0] == 0x12345678u) entries[0] = 0; return;
#endif
    // ---- C: every entry to its place in the tile-sorted order ----
    const uint16_t* cw = cntw + wave * NTp;
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        const uint32_t rg = rst[s].x, sb = rst[s].y;
        const int nx = (rg >> 16) & 3, ny = (rg >> 18) & 3;
        const int tx0 = rg & 0xff, ty0 = (rg >> 8) & 0xff;
#pragma unroll
        for (int j = 0; j < 4; j++) {
#ifdef EORB_TEST_RANK_ORDER_BUG
            const int dx = j & 1, dy = j >> 1;
#else
            const int dx = ((j & 1) - tx0) & 1, dy = ((j >> 1) - ty0) & 1;
#endif
            if (nx && ny && dx < nx && dy < ny) {
                const int t = (ty0 + dy) * TX + tx0 + dx;
                const uint32_t pos = (uint32_t)loff[t] + cw[t] + ((rk[s] >> (8 * j)) & 0xffu);
                sorted[pos] = (uint8_t)((sb >> (8 * (dy * 2 + dx))) & 0xffu);
                stile[pos] = (uint16_t)t;
            }
        }
    }
    __syncthreads();
#if defined(EORB_KO_SCAT) && EORB_KO_SCAT == 3
    if (rk[0] == 0x12345678u) entries[0] = 0; return;
#endif
    // ---- D: consecutive threads write consecutive entries of a run ----
    slot_entry* out = entries + (size_t)slice_ebase[cd.slice];
    const int E = loff[NT];
    for (int p = tid; p < E; p += NTHR) out[(size_t)(uint32_t)(gbase[stile[p]] + (uint32_t)p)] = (slot_entry)(kEntryTag | sorted[p]);     // (gbase holds base - loff mod 2^32)
}

// ---- K1c, with the ranks of the count pass (sl_count_lds_kernel<.., RANKS>): an 8-byte record per event { sensor index : 16, rank in the
// chunk's run of the event's tile of class j : 12 bits each }.  Nothing is counted here: the chunk's tile offsets are the prefix of
// the count pass's own counts (segcnt), an entry's place in the chunk's tile-sorted order is offset + rank.  Phases C and D as above.
template <int NW>
__global__ __launch_bounds__(64 * NW) void sl_scatter_pre_kernel(const uint64_t* __restrict__ rec, const ChunkDesc* __restrict__ chunks,
                                                                 const uint2* __restrict__ slot_tab, int nsrc, int TX, int NT, int chunk_cap,
                                                                 const int64_t* __restrict__ slice_ebase, const uint16_t* __restrict__ segcnt,
                                                                 const uint32_t* __restrict__ segbase, const uint32_t* __restrict__ tile_base,
                                                                 slot_entry* __restrict__ entries)
{
    extern __shared__ unsigned char sm2[];
    __shared__ uint32_t s_wsum[NW];
    constexpr int NTHR = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x;
    const ChunkDesc cd = chunks[chunk];
    const int NTp = (NT + 1) & ~1;
    uint32_t* gbase = (uint32_t*)sm2;                                 // NT: first entry of the tile's run in the global lists, minus loff
    uint16_t* stile = (uint16_t*)(gbase + NT);                        // chunk_cap * 4: tile of every slot of the sorted order
    uint16_t* loff = stile + (size_t)chunk_cap * 4;                   // NT + 1 (+ 1 pad)
    uint8_t* sorted = (uint8_t*)(loff + NTp + 2);                     // chunk_cap * 4: the entry bytes in tile-sorted order (tile and byte in ONE 32-bit word: 0.75 ms against 0.70)
    const int Q = (((cd.n + NW - 1) / NW) + 63) & ~63;
    const int S = Q >> 6;
    constexpr int SMAX = 4;
    // ---- A': the records and the slot table rows (all requests before the first use) ----
    uint64_t rr[SMAX]; uint2 rst[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        const int k = wave * Q + s * 64 + lane;
        rr[s] = (s < S && k < cd.n) ? rec[cd.start + k] : 0xffffull;
    }
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        const uint32_t row = (uint32_t)(rr[s] & 0xffffull);
        rst[s] = (row != 0xffffu && row < (uint32_t)nsrc) ? slot_tab[row] : make_uint2(0u, 0xffffffffu);
    }
    // ---- B': the chunk's tile offsets = exclusive scan of the count pass's counts over the tiles ----
    {
        const int per = (NT + NTHR - 1) / NTHR;
        const int t0 = tid * per, t1 = min(t0 + per, NT);
        const uint16_t* sc = segcnt + (size_t)chunk * NT;
        uint32_t mine = 0;
        for (int t = t0; t < t1; t++) mine += sc[t];
        uint32_t incl = (uint32_t)wave_incl_scan((int)mine);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w = 0; w < wave; w++) before += s_wsum[w];
        for (int t = t0; t < t1; t++) {
            const uint32_t v = sc[t];
            loff[t] = (uint16_t)before;
            gbase[t] = tile_base[(size_t)cd.slice * NT + t] + segbase[(size_t)chunk * NT + t] - before;
            before += v;
        }
        if (tid == NTHR - 1) loff[NT] = (uint16_t)before;
    }
    __syncthreads();
#if defined(EORB_KO_SCAT) && EORB_KO_SCAT == 2
    if (rst[0].x == 0x12345678u) entries[0] = 0; return;
#endif
    // ---- C: every entry to its place in the tile-sorted order ----
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        const uint32_t rg = rst[s].x, sb = rst[s].y;
        const int nx = (rg >> 16) & 3, ny = (rg >> 18) & 3;
        const int tx0 = rg & 0xff, ty0 = (rg >> 8) & 0xff;
        const uint64_t rk = rr[s] >> 16;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int dx = ((j & 1) - tx0) & 1, dy = ((j >> 1) - ty0) & 1;
            if (nx && ny && dx < nx && dy < ny) {
                const int t = (ty0 + dy) * TX + tx0 + dx;
                const uint32_t pos = (uint32_t)loff[t] + (uint32_t)((rk >> (12 * j)) & 0xfffull);
                sorted[pos] = (uint8_t)((sb >> (8 * (dy * 2 + dx))) & 0xffu);
                stile[pos] = (uint16_t)t;
            }
        }
    }
    __syncthreads();
#if defined(EORB_KO_SCAT) && EORB_KO_SCAT == 3
    if (rst[0].x == 0x12345678u) entries[0] = 0; return;
#endif
    // ---- D: consecutive threads write consecutive entries of a run ----
    slot_entry* out = entries + (size_t)slice_ebase[cd.slice];
    const int E = loff[NT];
    for (int p = tid; p < E; p += NTHR) out[(size_t)(uint32_t)(gbase[stile[p]] + (uint32_t)p)] = (slot_entry)(kEntryTag | sorted[p]);     // (gbase holds base - loff mod 2^32)
}

// Are the results of a wave's LDS atomic add handed out in lane order among the lanes that hit the same counter (32-bit words
// holding two 16-bit counters, some lanes inactive)?  bad = number of pairs out of order.  Run once per context with the shapes the
// scatter itself uses: 8 and 16 wavefronts per workgroup, every wavefront on its own counter row, plain LDS stores of the other
// wavefronts in flight between the atomics (phase A of the scatter overlaps phase C of nobody, but the count kernels' stores do).
__global__ __launch_bounds__(1024) void sl_rankcheck_kernel(unsigned long long* bad)
{
    __shared__ uint32_t cnt[16 * 64];                                 // 128 16-bit counters per wavefront
    __shared__ uint32_t noise[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long nbad = 0;
    uint32_t h = (uint32_t)(blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    for (int t = 0; t < 64; t++) {
        for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) cnt[i] = 0;
        __syncthreads();
        h = h * 1664525u + 1013904223u;
        const int spread = 1 << (1 + (t % 7));                       // 2 ... 128 distinct counters per wave
        const uint32_t a = ((h >> 9) % spread) + wave * 128;
        const bool act = ((h >> 3) & 7u) != 0u;
        uint32_t r = 0xffffffffu;
        noise[(threadIdx.x * 33 + t) & 1023] = h;                    // plain stores beside the atomics
        // four atomics in a row, as the scatter issues them (one per parity class): the order inside each must hold
        uint32_t rr[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            rr[q] = 0xffffffffu;
            if (act && ((h >> (12 + q)) & 1u)) { const uint32_t o = atomicAdd(&cnt[a >> 1], 1u << (16 * (a & 1))); rr[q] = (o >> (16 * (a & 1))) & 0xffffu; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            r = rr[q];
            for (int j = 0; j < 64; j++) {
                const uint32_t aj = __shfl(a, j, 64), rj = __shfl(r, j, 64);
                if (r != 0xffffffffu && rj != 0xffffffffu && j < lane && aj == a && !(rj < r)) nbad++;
            }
        }
        __syncthreads();
    }
    if (noise[threadIdx.x] == 0x12345u) nbad += 0;                    // (keeps the stores)
    if (nbad) atomicAdd(bad, nbad);
}

// ---- the gather's work plan ----
// (1) per tile position: its (slice, tile) lists sorted longest first (a list is a serial chain of adds: the long ones must start
//     early), each with its descriptor { slice, entries, list offset } so that a ticket costs the gather ONE load; the position's
//     total and its longest list
// A list of hot_min entries or more goes to sl_hot_kernel instead (rows in registers, a third of the time per entry): its descriptor
// { slice, tile, entries, list offset (2), tile x0, tile y0, byte offset of the tile's rows } lands in one of 16 buckets by length
// (bucket 0 = longest: the hot kernel takes them in that order), and the position's own list carries it with the top bit of the slice
// set, for the gather to skip.
constexpr int kHotBuckets = 16, kHotCap = 8192;
struct HotDesc { uint32_t slice, tile, cnt, off_lo, off_hi, tx0, ty0, rows_off; };
__global__ __launch_bounds__(256) void sl_plan_kernel(const uint32_t* __restrict__ tile_cnt, const uint32_t* __restrict__ tile_base,
                                                      const int64_t* __restrict__ slice_ebase, int B, int NT, int TX, uint32_t hot_min,
                                                      const uint32_t* __restrict__ nslots, const uint32_t* __restrict__ rowbase,
                                                      uint4* __restrict__ items, uint32_t* __restrict__ tile_w, uint32_t* __restrict__ tile_m,
                                                      uint32_t* __restrict__ ctr, uint32_t* __restrict__ hot_cnt, HotDesc* __restrict__ hot_items,
                                                      uint32_t hot_cap /* lists per bucket: kHotCap, less under the test hook */)
{
    extern __shared__ uint32_t pc[];                 // B counts | B (bucket << 16 | index in the block's share of the bucket)
    __shared__ uint32_t red[8];
    __shared__ uint32_t hb[kHotBuckets], hbase[kHotBuckets];
    uint32_t* hloc = pc + B;
    const int t = blockIdx.x, tid = threadIdx.x;
    const bool hot_ok = hot_min != 0u && nslots[t] <= (uint32_t)SL_HOT_NROWS;
    if (tid < kHotBuckets) hb[tid] = 0u;
    __syncthreads();
    for (int s = tid; s < B; s += blockDim.x) {
        const uint32_t c = tile_cnt[(size_t)s * NT + t];
        uint32_t v = c;
        if (hot_ok && c >= hot_min) {
            // quarter octaves above hot_min, longest first; the block's lists of a bucket are counted here, placed below
            const int lg = 31 - __clz(c), lg0 = 31 - __clz(hot_min);
            const int q = (lg - lg0) * 4 + (int)((c >> max(lg - 2, 0)) & 3u) - (int)((hot_min >> max(lg0 - 2, 0)) & 3u);
            const int b = kHotBuckets - 1 - min(max(q, 0), kHotBuckets - 1);
            hloc[s] = ((uint32_t)b << 16) | atomicAdd(&hb[b], 1u);
            v = c | 0x80000000u;
        }
        pc[s] = v;
    }
    __syncthreads();
    if (tid < kHotBuckets) hbase[tid] = hb[tid] ? atomicAdd(&hot_cnt[tid], hb[tid]) : 0u;      // (one atomic per bucket and position: sl_tasks_kernel caps the totals)
    __syncthreads();
    uint32_t sum = 0, mx = 0;
    for (int s = tid; s < B; s += blockDim.x) {
        uint32_t v = pc[s];
        if (v & 0x80000000u) {
            const uint32_t c = v & 0x7fffffffu, b = hloc[s] >> 16, k = hbase[b] + (hloc[s] & 0xffffu);
            if (k < hot_cap) {
                const uint64_t off = ((uint64_t)slice_ebase[s] + tile_base[(size_t)s * NT + t]) * sizeof(slot_entry);      // bytes
                HotDesc d; d.slice = (uint32_t)s; d.tile = (uint32_t)t; d.cnt = c; d.off_lo = (uint32_t)off; d.off_hi = (uint32_t)(off >> 32);
                d.tx0 = (uint32_t)(t % TX) * 8u; d.ty0 = (uint32_t)(t / TX) * 8u; d.rows_off = rowbase[t] * 256u;
                hot_items[(size_t)b * kHotCap + k] = d;
            } else { v = c; pc[s] = v; atomicAdd(&hot_cnt[33], 1u); }      // the bucket is full: the gather keeps the list (counted: eorb_debug_counter "slot_hot_overflow")
        }
        if (!(v & 0x80000000u)) { sum += v; mx = max(mx, v); }
    }
    __syncthreads();
    const bool sorted = B <= 2048;
    for (int s = tid; s < B; s += blockDim.x) {
        const uint32_t c = pc[s];
        int rank = s;
        if (sorted) { rank = 0; for (int j = 0; j < B; j++) { const uint32_t v = pc[j]; rank += (v > c || (v == c && j < s)) ? 1 : 0; } }
        const uint64_t off = ((uint64_t)slice_ebase[s] + tile_base[(size_t)s * NT + t]) * sizeof(slot_entry);          // bytes
        items[(size_t)t * B + rank] = make_uint4((uint32_t)s | (c & 0x80000000u), c & 0x7fffffffu, (uint32_t)off, (uint32_t)(off >> 32));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sum += (uint32_t)__shfl_xor((int)sum, d, 64); mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64)); }
    if ((tid & 63) == 0) { red[tid >> 6] = sum; red[4 + (tid >> 6)] = mx; }
    __syncthreads();
    if (tid == 0) {
        tile_w[t] = red[0] + red[1] + red[2] + red[3];
        tile_m[t] = max(max(red[4], red[5]), max(red[6], red[7]));
        ctr[t] = 0u;
    }
}
// (2) the workgroup tasks: tile positions ordered by their longest list, position t repeated n_t = 1 + its share of the G - NT spare
//     tasks by total entries (no more than its slices can occupy); unused tasks carry 0xffffffff
__global__ __launch_bounds__(1024) void sl_tasks_kernel(const uint32_t* __restrict__ tile_w, const uint32_t* __restrict__ tile_m, int NT, int G,
                                                        int max_per_tile, uint32_t* __restrict__ scratch /* 2 * NT */, uint32_t* __restrict__ task_tile, uint32_t* __restrict__ hot_cnt,
                                                        uint32_t hot_cap)
{
    if (threadIdx.x < kHotBuckets) hot_cnt[threadIdx.x] = min(hot_cnt[threadIdx.x], hot_cap);
    __shared__ unsigned long long wred[16];
    __shared__ uint32_t wsum[16], wsum2[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long tot = 0;
    for (int t = tid; t < NT; t += 1024) tot += tile_w[t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += (unsigned long long)__shfl_xor((long long)tot, d, 64);
    if (lane == 0) wred[wave] = tot;
    __syncthreads();
    tot = 0;
    for (int w = 0; w < 16; w++) tot += wred[w];
    const unsigned long long spare = (unsigned long long)max(G - NT, 0);
    uint32_t* nsorted = scratch; uint32_t* tsorted = scratch + NT;
    extern __shared__ uint32_t tm[];                 // NT: the longest lists (LDS copy for the rank loop)
    for (int t = tid; t < NT; t += 1024) tm[t] = tile_m[t];
    __syncthreads();
    for (int t = tid; t < NT; t += 1024) {
        const uint32_t m = tm[t];
        int rank = 0;
        for (int u = 0; u < NT; u++) { const uint32_t v = tm[u]; rank += (v > m || (v == m && u < t)) ? 1 : 0; }
        uint32_t n = 1u + (tot ? (uint32_t)((unsigned long long)tile_w[t] * spare / tot) : 0u);
        n = min(n, (uint32_t)max(max_per_tile, 1));
        nsorted[rank] = n; tsorted[rank] = (uint32_t)t;
    }
    __syncthreads();
    // first up to two tasks per position in that order (the longest chains of the heavy positions start in the first round on
    // different CUs), then the remaining ones: two exclusive scans over the ranks (consecutive ranks per thread)
    const int per = (NT + 1023) / 1024;
    const int r0 = tid * per, r1 = min(r0 + per, NT);
    uint32_t mine1 = 0, mine2 = 0;
    for (int r = r0; r < r1; r++) { const uint32_t n = nsorted[r], a = min(n, 2u); mine1 += a; mine2 += n - a; }
    const uint32_t incl1 = (uint32_t)wave_incl_scan((int)mine1), incl2 = (uint32_t)wave_incl_scan((int)mine2);
    if (lane == 63) { wsum[wave] = incl1; wsum2[wave] = incl2; }
    __syncthreads();
    uint32_t b1 = incl1 - mine1, b2 = incl2 - mine2, tot1 = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) { b1 += wsum[w]; b2 += wsum2[w]; } tot1 += wsum[w]; }
    b2 += tot1;
    for (int r = r0; r < r1; r++) {
        const uint32_t n = nsorted[r], a = min(n, 2u), t = tsorted[r];
        for (uint32_t k = 0; k < a; k++) if (b1 + k < (uint32_t)G) task_tile[b1 + k] = t;
        for (uint32_t k = 0; k < n - a; k++) if (b2 + k < (uint32_t)G) task_tile[b2 + k] = t;
        b1 += a; b2 += n - a;
    }
    const uint32_t before = b2;
    if (tid == 1023) for (uint32_t k = before; k < (uint32_t)G; k++) task_tile[k] = 0xffffffffu;
}

// ---- K2p ----
// One workgroup per task = a tile position (heavy positions get several tasks): the tile's rows -> LDS once; then every wavefront
// takes items of the position by ticket (longest first, one ticket at a time, requested when the item in hand has at most four
// blocks to go so that the round trip hides behind its adds).  A list is read 1024 entries at a time (one global_load_dwordx4 per
// lane, the next block requested before the current one is consumed); entries past the end of the list become the null slot (a row of
// zeros).  The inner loop walks one lane's 16 entries per iteration: v_readlane -> SGPR, v_perm_b32 builds the row's LDS address
// { slot, 4 * lane } per entry, ds_read_b32 three groups of four ahead of their adds (counted lgkmcnt), all inside ONE asm
// statement that drains its reads before it ends (no load is in flight across the statement's boundary).
// (Sharing a long list out by pixel quadrant was built and dropped: a 7x7 stamp centred inside an 8x8 tile reaches all four quadrants,
// so on the tiles that matter a quadrant's wave keeps 95 % of the entries.)
struct SlotGather {
    const uint32_t* task_tile; const uint4* items; const uint8_t* entries;      /* (byte pointer: the descriptors carry byte offsets) */
    const uint32_t* nslots; const uint32_t* rowbase; const float* rows; const uint32_t* tile_w; uint32_t* ctr;
    float* img; uint32_t* minmax_enc; int* info; int* status;
    int B, W, H, TX, NT, null_slot; uint32_t prio_ref;
    unsigned long long* trace;      // EORB_SLOT_TRACE builds: per wave { tile, start, end, entries, items } (wall clock, 100 MHz)
};

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
void sl_gather_kernel(SlotGather P)
{
    // [slot][64] floats; row null_slot = zeros.  The ONLY LDS object: offset 0
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef EORB_SLOT_TRACE
    const unsigned long long tr_t0 = wall_clock64(); unsigned long long tr_ent = 0, tr_items = 0, tr_kept = 0;
#endif
    const uint32_t task = P.task_tile[blockIdx.x];
    if (task == 0xffffffffu) return;
    const int tile = (int)task;
    const uint32_t wt = P.tile_w[tile];
    const int tx0 = (tile % P.TX) * kTile, ty0 = (tile / P.TX) * kTile;
    const int px = tx0 + (lane & 7), py = ty0 + (lane >> 3);
    const bool inimg = px < P.W && py < P.H;
    if ((uint32_t)(uintptr_t)lds != 0u) { if (tid == 0) { atomicOr(&P.info[3], 1); atomicOr(P.status, 256); } return; }      // (reported by eorb_sync / the host entry point)
    if (wt) {
        const int ns = (int)P.nslots[tile];
        const int n4 = ns * 16;
        const float4* src = (const float4*)(P.rows + (size_t)P.rowbase[tile] * 64);
        float4* dst = (float4*)lds;
        for (int i = tid; i < n4; i += blockDim.x) dst[i] = src[i];
        if (tid < 16) dst[P.null_slot * 16 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const uint32_t nullword = (uint32_t)P.null_slot * 0x00010001u;
    uint32_t* const ctr = P.ctr + tile;
    // v_perm_b32 selectors: LDS byte address of an entry's row for this lane = { 0, 0, slot byte of the entry (byte 0 / 2 of the dword), 4 * lane }
    const uint32_t lane4 = (uint32_t)lane * 4u;
    uint32_t sel0 = 0x0c0c0400u, sel1 = 0x0c0c0600u;
    asm volatile("" : "+v"(sel0), "+v"(sel1));
    const int nit = P.B;
    const uint4* const items = P.items + (size_t)tile * P.B;
    auto ticket = [&]() { int t = 0; if (lane == 0) t = (int)atomicAdd(ctr, 1u); return t; };         // lane 0 holds the value
    auto uni = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto item_of = [&](int tkv) { return (int)min((uint32_t)__builtin_amdgcn_readfirstlane(tkv), (uint32_t)nit); };
    int k = item_of(ticket());
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (k < nit) d = items[k];
    uint4 En = make_uint4(0u, 0u, 0u, 0u);
    bool have_first = false;
    while (k < nit) {
        const bool hot = (uni(d.x) >> 31) != 0u;                           // taken by sl_hot_kernel: nothing to do here
        const int s = (int)(uni(d.x) & 0x7fffffffu);
        const uint32_t cnt = hot ? 0u : uni(d.y);
        const uint4* const list = (const uint4*)(P.entries + (((uint64_t)uni(d.w) << 32) | uni(d.z)));
        // a long list is a serial chain of adds: its wave goes first at the issue arbiter
        if (cnt >= P.prio_ref) __builtin_amdgcn_s_setprio(3);
        else if (cnt >= P.prio_ref / 4u) __builtin_amdgcn_s_setprio(2);
        else if (cnt >= P.prio_ref / 16u) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        float acc = 0.0f;
        const int nblk = (int)((cnt + 511u) >> 9);           // a block = 64 lanes x 8 entries
        if (!have_first && nblk) En = list[lane];
        have_first = false;
        int tk = 0; int stage_t = 0;                   // 0: no ticket yet, 1: ticket requested, 2: descriptor requested
        uint4 dn = make_uint4(0u, 0u, 0u, 0u); int kn = nit;
        for (int b = 0; b < nblk; b++) {
            uint4 E = En;
            // the next item: ticket when at most four blocks are left, its descriptor one block later, its first block with the last one
            if (stage_t == 1) { kn = item_of(tk); if (kn < nit) dn = items[kn]; stage_t = 2; }
            else if (stage_t == 0 && nblk - b <= 4) { tk = ticket(); stage_t = 1; }
            if (b + 1 < nblk) En = list[(size_t)(b + 1) * 64 + lane];
            else if (stage_t == 2 && kn < nit && uni(dn.y)) {
                En = ((const uint4*)(P.entries + (((uint64_t)uni(dn.w) << 32) | uni(dn.z))))[lane];
                have_first = true;
            }
            const int rem = (int)cnt - b * 512;
            if (rem < 512) {
                // entries past the end of the list -> the null slot
                auto fix = [&](uint32_t w, int dd) {
                    const int nv = rem - lane * 8 - dd * 2;                   // valid entries of this dword
                    const uint32_t keep = nv >= 2 ? 0xffffffffu : (nv <= 0 ? 0u : 0xffffu);
                    return (w & keep) | (nullword & ~keep);
                };
                E.x = fix(E.x, 0); E.y = fix(E.y, 1); E.z = fix(E.z, 2); E.w = fix(E.w, 3);
            }
            const int lane_end = (min(64, (rem + 7) >> 3) + 1) & ~1;     // two lanes (8 entries each) per iteration; an odd count takes one lane of null entries along
#ifdef EORB_SLOT_TRACE
            tr_kept += (unsigned long long)lane_end * 8;
#endif
            {
            float r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
            int sl, sl1, se, sf;
            // two dwords = four entries: addresses by v_perm_b32, reads into the address registers, adds of the group read three groups ago
#define SL_GROUP(EV, EW, SL, RA, RB, RC, RD, AA, AB, AC, AD) \
            "s_waitcnt lgkmcnt(8)\n" \
            "v_readlane_b32 %[se], %[" EV "], %[" SL "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AA "]\n" \
            "v_readlane_b32 %[sf], %[" EW "], %[" SL "]\n" \
            "v_perm_b32 %[" RA "], %[se], %[l4], %[q0]\n" \
            "v_perm_b32 %[" RB "], %[se], %[l4], %[q1]\n" \
            "ds_read_b32 %[" RA "], %[" RA "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AB "]\n" \
            "v_perm_b32 %[" RC "], %[sf], %[l4], %[q0]\n" \
            "ds_read_b32 %[" RB "], %[" RB "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AC "]\n" \
            "v_perm_b32 %[" RD "], %[sf], %[l4], %[q1]\n" \
            "ds_read_b32 %[" RC "], %[" RC "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AD "]\n" \
            "ds_read_b32 %[" RD "], %[" RD "]\n"
            asm volatile(
                "v_mov_b32 %[r4], 0\n v_mov_b32 %[r5], 0\n v_mov_b32 %[r6], 0\n v_mov_b32 %[r7], 0\n"
                "v_mov_b32 %[r8], 0\n v_mov_b32 %[r9], 0\n v_mov_b32 %[r10], 0\n v_mov_b32 %[r11], 0\n"
                "v_mov_b32 %[r12], 0\n v_mov_b32 %[r13], 0\n v_mov_b32 %[r14], 0\n v_mov_b32 %[r15], 0\n"
                "s_mov_b32 %[sl], 0\n"
                "1:\n"
                "s_add_u32 %[sl1], %[sl], 1\n"
                SL_GROUP("e0", "e1", "sl", "r0", "r1", "r2", "r3", "r4", "r5", "r6", "r7")
                SL_GROUP("e2", "e3", "sl", "r4", "r5", "r6", "r7", "r8", "r9", "r10", "r11")
                SL_GROUP("e0", "e1", "sl1", "r8", "r9", "r10", "r11", "r12", "r13", "r14", "r15")
                SL_GROUP("e2", "e3", "sl1", "r12", "r13", "r14", "r15", "r0", "r1", "r2", "r3")
                "s_add_u32 %[sl], %[sl], 2\n"
                "s_cmp_lt_u32 %[sl], %[lend]\n"
                "s_cbranch_scc1 1b\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_add_f32 %[acc], %[acc], %[r4]\n v_add_f32 %[acc], %[acc], %[r5]\n v_add_f32 %[acc], %[acc], %[r6]\n v_add_f32 %[acc], %[acc], %[r7]\n"
                "v_add_f32 %[acc], %[acc], %[r8]\n v_add_f32 %[acc], %[acc], %[r9]\n v_add_f32 %[acc], %[acc], %[r10]\n v_add_f32 %[acc], %[acc], %[r11]\n"
                "v_add_f32 %[acc], %[acc], %[r12]\n v_add_f32 %[acc], %[acc], %[r13]\n v_add_f32 %[acc], %[acc], %[r14]\n v_add_f32 %[acc], %[acc], %[r15]\n"
                : [acc] "+v"(acc), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5),
                  [r6] "=&v"(r6), [r7] "=&v"(r7), [r8] "=&v"(r8), [r9] "=&v"(r9), [r10] "=&v"(r10), [r11] "=&v"(r11),
                  [r12] "=&v"(r12), [r13] "=&v"(r13), [r14] "=&v"(r14), [r15] "=&v"(r15),
                  [sl] "=&s"(sl), [sl1] "=&s"(sl1), [se] "=&s"(se), [sf] "=&s"(sf)
                : [e0] "v"(E.x), [e1] "v"(E.y), [e2] "v"(E.z), [e3] "v"(E.w), [lend] "s"(lane_end), [l4] "v"(lane4),
                  [q0] "v"(sel0), [q1] "v"(sel1), [ldsp] "v"(lds)
                : "scc", "memory");
#undef SL_GROUP
            }
        }
        // the slice's pixels of this tile; an empty list offers nothing to the running extremes (max stays -1e6:
        // resolveMinMaxVals :32-39)
        if (inimg && !hot) P.img[(size_t)s * P.W * P.H + (size_t)py * P.W + px] = acc;
        if (cnt) {
            // every increment is >= 0: the running maximum is the largest final value of a visited tile, the minimum stays 0
            float vmax = inimg ? acc : -1000000.0f;
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, dd, 64));
            if (lane == 0) atomicMax(&P.minmax_enc[s * 2 + 1], enc_f32(vmax));
        }
#ifdef EORB_SLOT_TRACE
        tr_ent += cnt; tr_items++;
#endif
        // next item
        if (stage_t == 0) { tk = ticket(); stage_t = 1; }
        if (stage_t == 1) { kn = item_of(tk); if (kn < nit) dn = items[kn]; }
        k = kn; d = dn;
    }
#ifdef EORB_SLOT_TRACE
    if (lane == 0 && P.trace) {
        unsigned long long* r = P.trace + ((size_t)blockIdx.x * 16 + (tid >> 6)) * 6;
        r[0] = (unsigned long long)tile; r[1] = tr_t0; r[2] = wall_clock64(); r[3] = tr_ent; r[4] = tr_items; r[5] = tr_kept;
    }
#endif
}

// ---- K2h: the long lists.  One wavefront per item at a time (tickets over the 16 length buckets, longest first): the tile
// position's rows are loaded into VGPRs v0..v239 (lane = pixel, exactly one register per row), and every 16-bit entry is one scalar
// move to M0 (0x1000 | slot = "SRC0 relative, index slot" in the VGPR index mode of this ISA) + one v_add_f32 acc, v[M0], acc:
// 8-9 cycles per entry and wavefront against 29 for the LDS form (tools/mb/gpr_idx.hip), two wavefronts per SIMD = all of its
// registers.  Entries arrive by scalar loads, two 32-SGPR buffers, behind a vector-load prefetch into the L2 (tools/gen_sl_hot.py,
// which generates the body and explains the schedule).
__global__ __launch_bounds__(64) void sl_hot_kernel(const uint32_t* __restrict__ hcnt, uint32_t* __restrict__ ticket, const HotDesc* __restrict__ items,
                                                    int hcap, const float* __restrict__ rows, const slot_entry* __restrict__ entries,
                                                    float* __restrict__ img, uint32_t* __restrict__ mm, int W, int H)
{
    const uint32_t rows_lo = (uint32_t)(uintptr_t)rows, rows_hi = (uint32_t)((uintptr_t)rows >> 32);
    const uint32_t ent_lo = (uint32_t)(uintptr_t)entries, ent_hi = (uint32_t)((uintptr_t)entries >> 32);
    const uint32_t img_lo = (uint32_t)(uintptr_t)img, img_hi = (uint32_t)((uintptr_t)img >> 32);
    asm volatile(SL_HOT_ASM
                 :
                 : [hcnt] "s"(hcnt), [ticket] "s"(ticket), [items] "s"(items), [hcap] "s"(hcap), [rows_lo] "s"(rows_lo), [rows_hi] "s"(rows_hi),
                   [ent_lo] "s"(ent_lo), [ent_hi] "s"(ent_hi), [img_lo] "s"(img_lo), [img_hi] "s"(img_hi), [mm] "s"(mm), [W] "s"(W), [H] "s"(H)
                 : SL_HOT_CLOBBERS);
}

// ---- host ----
// Two halves, so that a caller that synchronises anyway (the float bulk path waits for its position count) can put the slot
// assignment in front of ITS wait: launch = assignment + row bases (+ the once-per-context rank check), info read back into
// c->sl_hinfo by an asynchronous copy; finish = after the stream was waited for: the rows, sl_ok.
int ev_slots_prepare_launch(eorb_ctx* c, int W, int H, int h, int TX, int TY)
{
    c->sl_ok = 0; c->sl_launched = 0;
    if (h < 1 || h > 4 || TX >= 256 || TY >= 256) return EORB_OK;
    const int nsrc = c->lut_w * c->lut_h, NT = TX * TY;
    int rc;
    const size_t geo_off = sizeof(uint2) * (size_t)nsrc;                // slot_tab | slot_geo (16 bits per sensor pixel, + pad)
    if ((rc = ensure(c, c->sl_tab, geo_off + sizeof(uint16_t) * ((size_t)nsrc + 2)))) return rc;
    // nslots | rowbase | tile_w | ctr | (spare) (NT each) | info (4 ints) | rank check (u64)
    if ((rc = ensure(c, c->sl_tile, sizeof(uint32_t) * (5 * (size_t)NT + 8)))) return rc;
    uint32_t* d_nslots = (uint32_t*)c->sl_tile.p;
    uint32_t* d_rowbase = d_nslots + NT;
    int* d_info = (int*)(d_nslots + 5 * (size_t)NT);
    c->sl_info_off = sizeof(uint32_t) * 5 * (size_t)NT;
    EORB_HIP(c, hipMemsetAsync(c->sl_tile.p, 0, sizeof(uint32_t) * (5 * (size_t)NT + 8), c->stream));
    sl_assign_kernel<<<(nsrc + 256) / 256, 256, 0, c->stream>>>((const uint32_t*)c->src_info.p, nsrc, W, H, h, TX, d_nslots, (uint2*)c->sl_tab.p,
                                                                (uint16_t*)((char*)c->sl_tab.p + geo_off), d_info);
    sl_rowbase_kernel<<<1, 1024, 0, c->stream>>>(d_nslots, NT, d_rowbase, d_info);
    EORB_LAUNCH_CHECK(c, "slot table kernels");
    if (c->sl_rank_ok < 0) {
        // once per context: may the scatter take its stable ranks from LDS atomics? (see sl_scatter_rank_kernel)
        unsigned long long* d_bad = (unsigned long long*)(d_info + 4);
        sl_rankcheck_kernel<<<128, 512, 0, c->stream>>>(d_bad);            // 8 wavefronts: the 2 048-event chunks
        sl_rankcheck_kernel<<<128, 1024, 0, c->stream>>>(d_bad);           // 16 wavefronts: the 4 096-event chunks
    }
    int* rb = readback_buf(c);
    if (!rb) return set_err(c, EORB_E_HIP, "pinned alloc failed");
    EORB_HIP(c, hipMemcpyAsync(rb + 8, d_info, sizeof(c->sl_hinfo), hipMemcpyDeviceToHost, c->stream));
    c->sl_launched = 1;
    return EORB_OK;
}

int ev_slots_prepare_finish(eorb_ctx* c, int W, int H, int h, int TX, int TY, const float* d_stamps, int stamp_stride, int SWP, float two_sig2, float norm)
{
    if (!c->sl_launched) return EORB_OK;
    c->sl_launched = 0;
    const int nsrc = c->lut_w * c->lut_h, NT = TX * TY;
    memcpy(c->sl_hinfo, readback_buf(c) + 8, sizeof(c->sl_hinfo));
    const int* hinfo = c->sl_hinfo;
    int rc;
    if (c->sl_rank_ok < 0) c->sl_rank_ok = (hinfo[4] == 0 && hinfo[5] == 0) ? 1 : 0;
    if (hinfo[2] || hinfo[1] >= (int)kNoSlot || hinfo[0] <= 0) return EORB_OK;      // a tile with more than 254 slots: the batch pipeline serves these maps
    if ((rc = ensure(c, c->sl_rows, sizeof(float) * 64 * (size_t)hinfo[0] + 4096 + 256 * 256))) return rc;      // + slack: sl_hot_kernel loads 240 rows whatever the tile holds
    const uint32_t* d_rowbase = (const uint32_t*)c->sl_tile.p + NT;
    sl_rows_kernel<<<(nsrc + 3) / 4, 256, 0, c->stream>>>((const uint32_t*)c->src_info.p, (const uint2*)c->sl_tab.p, nsrc, W, H, h, TX, d_stamps,
                                                            stamp_stride, SWP, (const float2*)c->lut.p, two_sig2, norm, d_rowbase, (float*)c->sl_rows.p);
    EORB_LAUNCH_CHECK(c, "sl_rows_kernel");
    c->sl_null = hinfo[1];
    c->sl_ok = 1;
    return EORB_OK;
}

// tables of the current maps / sigma (called from ev_raw_tables when they change); c->sl_ok = 1 when the slot form can run
int ev_slots_prepare(eorb_ctx* c, int W, int H, int h, int TX, int TY, const float* d_stamps, int stamp_stride, int SWP, float two_sig2, float norm)
{
    int rc;
    if (!c->sl_launched) {
        if ((rc = ev_slots_prepare_launch(c, W, H, h, TX, TY))) return rc;
        if (!c->sl_launched) return EORB_OK;
        EORB_HIP(c, hipStreamSynchronize(c->stream));
    }
    return ev_slots_prepare_finish(c, W, H, h, TX, TY, d_stamps, stamp_stride, SWP, two_sig2, norm);
}

// dynamic-LDS opt-in of one kernel instantiation, once per CONTEXT (= per device: a second context may sit on another GPU)
static int sl_optin(eorb_ctx* c, int bit, const void* fn, int bytes)
{
    if (c->sl_attr & (1u << bit)) return EORB_OK;
    EORB_HIP(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    c->sl_attr |= 1u << bit;
    return EORB_OK;
}
static int sl_ncu(eorb_ctx* c)
{
    if (!c->ncu) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, c->device) == hipSuccess) c->ncu = pr.multiProcessorCount; if (c->ncu <= 0) c->ncu = 256; }
    return c->ncu;
}

constexpr int kSlotDeclined = 1;       // ev_slots_accumulate: the batch's shape does not fit this form, nothing was launched

// The scatter a batch will run -- decided before anything is launched: the preferred chunk size first, then shorter chunks; for each
// the rank form (its own LDS footprint, <= 160 KB with the opt-in) where the device check allows it, else the ballot form (64 KB).
// false: no form fits (e.g. VGA-class sensors: 4 800 tiles x 8 per-wave counter rows alone are 77 KB) and the caller falls back to
// the batch pipeline, whose first scatter form needs 4 bytes of LDS per tile only.
struct SlotScatterChoice { int chunk, waves; bool rank; size_t lds; };
static size_t sl_lds_rank(int NT, int chunk, int waves) { const int NTp = (NT + 1) & ~1; return ((size_t)NT * 4 + (size_t)chunk * 4 * 2 + (size_t)waves * NTp * 2 + (size_t)(NTp + 2) * 2 + (size_t)chunk * 4 + 15) & ~(size_t)15; }
static size_t sl_lds_pre(int NT, int chunk) { const int NTp = (NT + 1) & ~1; return ((size_t)NT * 4 + (size_t)chunk * 4 * 2 + (size_t)(NTp + 2) * 2 + (size_t)chunk * 4 + 15) & ~(size_t)15; }
static size_t sl_lds_ballot(int NT, int chunk) { const int NTp = (NT + 1) & ~1; return ((size_t)chunk * 4 + (size_t)chunk * 2 + (size_t)chunk * 4 * 2 + (size_t)kSlotScatWaves * NTp * 2 + (size_t)(NTp + 2) * 2 + (size_t)NT * 4 + 15) & ~(size_t)15; }
static bool sl_choose_scatter(const eorb_ctx* c, int NT, int64_t per_slice, bool rank_allowed, SlotScatterChoice* out)
{
    static const int chunk_env = [] { const char* e = getenv("EORB_SLOT_CHUNK"); const int v = e ? atoi(e) : 0; return (v == 256 || v == 1024 || v == 2048 || v == 4096) ? v : 0; }();
    // chunks of 4 096 events (16 wavefronts per scatter workgroup) where the rank scatter runs and the slices are long: per-chunk work
    // (the tiles' list bases, the prefix over the waves, the count rows) is shared by twice the events
    // (128 x 1 Mev: binning 1.46-1.51 ms with 2 048, 1.41-1.44 with 4 096)
    const int pref = chunk_env ? chunk_env : (per_slice >= (int64_t)1 << 18 ? 4096 : (per_slice >= (int64_t)1 << 17 ? 2048 : (per_slice >= (int64_t)1 << 14 ? 1024 : 256)));
    static const int sizes[4] = {4096, 2048, 1024, 256};
    for (int i = 0; i < 4; i++) {
        const int chunk = sizes[i];
        if (chunk > pref) continue;
        if (rank_allowed) {
            const int waves = chunk == 4096 ? 16 : kSlotScatWaves;
            const size_t l = sl_lds_rank(NT, chunk, waves);
            // (two or more workgroups per CU: the phases of different chunks overlap; one 130 KB workgroup per CU still beats falling back)
            if (l <= (size_t)(chunk == 4096 ? 79 : 158) * 1024) { *out = {chunk, waves, true, l}; return true; }
        }
        if (chunk <= 2048) {
            const size_t l = sl_lds_ballot(NT, chunk);
            if (l <= 64 * 1024) { *out = {chunk, kSlotScatWaves, false, l}; return true; }
        }
    }
    (void)c;
    return false;
}

// The streams and events of the slot form, made once per context, all or nothing: P = the gather's plan (single-block kernels beside
// the scatter), H = the register-row kernel (high priority: its lists are the launch's longest chains), G = the LDS gather of a
// half batch while the main stream bins the next half.
static int sl_streams(eorb_ctx* c)
{
    if (c->sl_side) return EORB_OK;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t st[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[eorb_ctx::kSlotEvents];
    for (auto& e : ev) e = nullptr;
    bool ok = hipStreamCreateWithPriority(&st[0], hipStreamNonBlocking, hi) == hipSuccess && hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&st[2], hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < eorb_ctx::kSlotEvents; i++) ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {                                       // a half-made set must not be used by a later call
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        for (auto& q : st) if (q) (void)hipStreamDestroy(q);
        return set_err(c, EORB_E_HIP, "slot form: streams / events");
    }
    c->sl_side = st[0]; c->sl_pstream = st[1]; c->sl_gstream = st[2];
    for (int i = 0; i < eorb_ctx::kSlotEvents; i++) c->sl_ev[i] = ev[i];
    return EORB_OK;
}

// count -> scan -> scatter -> plan -> gather for the B slices of one PART of a batch (the whole batch, or one of its halves), on the
// part's own workspaces.  Streams: binning on the context's stream; the plan on P; the long lists on H; the LDS gather on G when the
// batch runs in halves (so that the next half's binning, HBM- and LDS-atomic-bound, runs under it), else on the context's stream.
static int slots_part(eorb_ctx* c, eorb_ctx::SlotWS& ws, int part, int nparts, const void* d_events, int stride_in, const int64_t* h_offsets, int B,
                      int W, int H, int TX, int TY, float* d_f32, uint32_t* d_minmax_enc, const SlotScatterChoice& sc, const SlotDict* dict)
{
    // with a position dictionary the count pass reads the float events and writes the hashed records every later pass reads
    const int stride = dict ? (dict->rec2 ? 2 : -4) : stride_in;
    const int NT = TX * TY;
    const int64_t nev = h_offsets[B] - h_offsets[0];
    const int chunk = sc.chunk;
    hipStream_t M = c->stream, P = c->sl_pstream, Hs = c->sl_side, G = nparts > 1 ? c->sl_gstream : c->stream;
    hipEvent_t* E = c->sl_ev + 5 * part;             // fork (scan done), plan, scat, hot done, gather done
    // up to 256 slices: the descriptors are made on the device from the offsets (sl_chunks_kernel); more: on the host, one copy
    const bool on_dev = B <= kOffsetsInArg;
    std::vector<ChunkDesc> cds;
    std::vector<int> slice_c0(on_dev ? 0 : B + 1);
    std::vector<int64_t> slice_eb(on_dev ? 0 : B + 1);
    int64_t eb = 0, nch = 0;
    for (int b = 0; b < B; b++) {
        if (!on_dev) slice_c0[b] = (int)nch;
        const int64_t s = h_offsets[b], e = h_offsets[b + 1];
        if (!on_dev) slice_eb[b] = eb;
        eb = (eb + (e - s) * 4 + (int64_t)NT * 16 + 15) & ~(int64_t)15;   // <= 4 entries per event, every list rounded up to 16
        nch += (e - s + chunk - 1) / chunk;
        if (!on_dev)
            for (int64_t k = s; k < e; k += chunk) {
                ChunkDesc cd; cd.start = k; cd.n = (int32_t)std::min<int64_t>(chunk, e - k); cd.slice = b;
                cds.push_back(cd);
            }
    }
    if (nch >= (int64_t)1 << 31) return set_err(c, EORB_E_CAPACITY, "ev_accumulate: too many chunks");
    if (!on_dev) { slice_c0[B] = (int)nch; slice_eb[B] = eb; }
    const int nchunks = (int)nch;
    const size_t cd_bytes = sizeof(ChunkDesc) * (size_t)std::max(nchunks, 1);
    const size_t sc_bytes = (sizeof(int) * (size_t)(B + 1) + 7) & ~(size_t)7;
    const size_t eb_bytes = sizeof(int64_t) * (size_t)B;
    const int nb = B * NT;
    c->sl_last_nb[part] = nb;
    int rc;
    if ((rc = ensure(c, ws.chunks, cd_bytes + sc_bytes + eb_bytes))) return rc;
    const size_t cnt_bytes = (sizeof(uint16_t) * (size_t)std::max(nchunks, 1) * NT + 15) & ~(size_t)15;
    if ((rc = ensure(c, ws.segoff, cnt_bytes + sizeof(uint32_t) * (size_t)std::max(nchunks, 1) * NT))) return rc;
    if ((rc = ensure(c, ws.entries, sizeof(slot_entry) * (size_t)eb + 8192))) return rc;      // + slack: the gather requests blocks past a list's end
    if ((rc = ensure(c, ws.tile_order, sizeof(uint32_t) * 2 * (size_t)nb))) return rc;
    if (on_dev) {
        SliceOffsets so;
        for (int b = 0; b <= B; b++) so.off[b] = h_offsets[b];
        sl_chunks_kernel<<<std::max(1, (nchunks + 255) / 256), 256, 0, M>>>(so, B, chunk == 4096 ? 12 : (chunk == 2048 ? 11 : (chunk == 1024 ? 10 : 8)), NT, nchunks, (ChunkDesc*)ws.chunks.p, (int*)((char*)ws.chunks.p + cd_bytes),
                                                                                  (int64_t*)((char*)ws.chunks.p + cd_bytes + sc_bytes));
    }
    else {
        char* hp = (char*)pinned(c, cd_bytes + sc_bytes + eb_bytes);
        if (!hp) return set_err(c, EORB_E_HIP, "pinned alloc failed");
        if (nchunks) memcpy(hp, cds.data(), sizeof(ChunkDesc) * nchunks);
        memcpy(hp + cd_bytes, slice_c0.data(), sizeof(int) * (size_t)(B + 1));
        memcpy(hp + cd_bytes + sc_bytes, slice_eb.data(), eb_bytes);
        EORB_HIP(c, hipMemcpyAsync(ws.chunks.p, hp, cd_bytes + sc_bytes + eb_bytes, hipMemcpyHostToDevice, M));
        pinned_commit(c);
    }
    const ChunkDesc* d_chunks = (const ChunkDesc*)ws.chunks.p;
    const int* d_slice_c0 = (const int*)((char*)ws.chunks.p + cd_bytes);
    const int64_t* d_slice_eb = (const int64_t*)((char*)ws.chunks.p + cd_bytes + sc_bytes);
    uint16_t* d_segcnt = (uint16_t*)ws.segoff.p;
    uint32_t* d_segbase = (uint32_t*)((char*)ws.segoff.p + cnt_bytes);
    uint32_t* d_tile_cnt = (uint32_t*)ws.tile_order.p;
    uint32_t* d_tile_base = d_tile_cnt + nb;
    uint32_t* d_nslots = (uint32_t*)c->sl_tile.p;
    uint32_t* d_rowbase = d_nslots + NT;
    int* d_info = (int*)(d_nslots + 5 * (size_t)NT);
    const eorb_raw_event* d_ev = dict ? (const eorb_raw_event*)dict->rec : (const eorb_raw_event*)d_events;
    int scat_stride = stride;
    bool prerank = false;                             // the count pass left ranked 8-byte records: sl_scatter_pre_kernel
    const uint2* d_tab = (const uint2*)c->sl_tab.p;
    const int ncu = sl_ncu(c);
    const int NTp = (NT + 1) & ~1;
    // ---- the gather's plan: sizes first (its buffers are allocated before anything of the part is launched) ----
    const size_t lds_g = (size_t)(c->sl_null + 1) * 256;
    const int wg_per_cu = std::max(1, (int)((160 * 1024) / lds_g));
    static const int nw_env = [] { const char* e = getenv("EORB_SLOT_WAVES"); return e ? atoi(e) : 0; }();
    static const int ns_env = [] { const char* e = getenv("EORB_SLOT_ROUNDS"); return e ? atoi(e) : 0; }();
    // four wavefronts per SIMD saturate the vector ALUs and leave every list about full single-wave speed (measured: 8 per SIMD
    // process the same entries per second, each list at half the pace)
    int nw = std::min(16, std::max(1, 16 / wg_per_cu));
    if (nw_env >= 1 && nw_env <= 16) nw = nw_env;
    nw = std::min(nw, std::max(1, B));
    // one task per tile position plus a few rounds of spare ones shared out by weight; a position never gets more wavefronts than slices
    const int rounds = ns_env >= 1 ? ns_env : 4;
    const int Gt = NT + rounds * ncu * wg_per_cu;
    const int max_per_tile = (B + nw - 1) / nw;
    // items | task table | longest lists | scratch (2 NT) | tile weights | tickets
    if ((rc = ensure(c, ws.plan, sizeof(uint4) * (size_t)nb + sizeof(uint32_t) * ((size_t)Gt + 5 * (size_t)NT)))) return rc;
    uint4* d_items = (uint4*)ws.plan.p;
    uint32_t* d_task = (uint32_t*)(d_items + nb);
    uint32_t* d_tile_m = d_task + Gt;
    uint32_t* d_scr = d_tile_m + NT;
    uint32_t* d_tile_w = d_scr + 2 * (size_t)NT;
    uint32_t* d_ctr = d_tile_w + NT;
    const uint32_t prio_ref = (uint32_t)std::min<int64_t>(std::max<int64_t>(4096, nev / 2000), 0x7fffffff);   // lists this long go first at the issue arbiter
    // lists of a few thousand entries or more go to the register-row kernel (when every tile's rows fit its 240 registers): 13
    // cycles per entry instead of 29, which shortens the launch's longest chains AND moves more entries per second; shorter lists
    // would not repay the 240 row loads per list (measured at 128 x 1 Mev: threshold 4 096 ... 8 192 1.40-1.45 ms, 16 384 1.53, none 2.44)
    static const long long hot_env = [] { const char* e = getenv("EORB_SLOT_HOT_MIN"); return e ? atoll(e) : -1ll; }();
    uint32_t hot_min = c->sl_null <= SL_HOT_NROWS ? (uint32_t)std::min<int64_t>(std::max<int64_t>(4096, nev * nparts / 16000), 0x7fffffff) : 0u;
    const long long hot_over = c->dbg_slot_hot_min >= 0 ? c->dbg_slot_hot_min : hot_env;
    // (never below 64: the register-row kernel's tail load reads the 64 bytes that END at the list's end; 0 = that kernel off)
    if (hot_over >= 0) hot_min = (c->sl_null <= SL_HOT_NROWS && hot_over > 0) ? (uint32_t)std::min<long long>(std::max<long long>(hot_over, 64), 0x7fffffff) : 0u;
    const uint32_t hot_cap = c->dbg_slot_hot_cap > 0 ? (uint32_t)std::min(c->dbg_slot_hot_cap, kHotCap) : (uint32_t)kHotCap;
    if ((rc = ensure(c, ws.hot, sizeof(HotDesc) * (size_t)kHotBuckets * kHotCap + 256))) return rc;
    uint32_t* d_hot_cnt = (uint32_t*)ws.hot.p;                           // 16 bucket counts | ticket (at word 32) | overflow count (33) | descriptors (from byte 256)
    HotDesc* d_hot_items = (HotDesc*)((char*)ws.hot.p + 256);
    {
        std::unique_ptr<ProfScope> ps(new ProfScope(c, "ev_count"));      // (one scope per kernel of the binning: count, scan, scatter)
        const size_t lds = sizeof(uint32_t) * (size_t)NT;
        // the tile ranges of all sensor pixels + one set of counters per wavefront in the LDS of one workgroup per CU?
        const size_t nsrc = (size_t)c->lut_w * (size_t)c->lut_h;
        const size_t lds_c = 4 * ((nsrc + 2) / 2) + (size_t)kCountWaves * NTp * 2;
        static const int cl_env = [] { const char* e = getenv("EORB_SLOT_COUNT_LDS"); return e ? atoi(e) : 1; }();
        if (dict && !(TX <= 127 && TY <= 127 && lds_c <= 159 * 1024)) return set_err(c, EORB_E_CAPACITY, "slot form: the position dictionary needs the LDS count pass");
        if (nchunks && dict) {
            const uint16_t* d_geo = (const uint16_t*)((const char*)c->sl_tab.p + sizeof(uint2) * nsrc);
            const int g = std::min(ncu, (nchunks + kCountWaves - 1) / kCountWaves);
            static const int pre_env_d = [] { const char* e = getenv("EORB_SLOT_PRERANK"); return e ? atoi(e) : 1; }();
            const int pre_on_d = c->dbg_slot_prerank >= 0 ? c->dbg_slot_prerank : pre_env_d;
            if (pre_on_d && dict->rec2 && sc.rank && sl_lds_pre(NT, chunk) <= 158 * 1024) {
                // (the dense ids are 16-bit: the ranked 8-byte record serves the dictionary path too)
                if ((rc = ensure(c, ws.rec16, sizeof(uint64_t) * (size_t)std::max<int64_t>(h_offsets[B], 1)))) return rc;
                prerank = true;
                if ((rc = sl_optin(c, 21, (const void*)sl_count_lds_kernel<16, true, true>, 159 * 1024))) return rc;
                sl_count_lds_kernel<16, true, true><<<g, 64 * kCountWaves, lds_c, M>>>((const eorb_raw_event*)d_events, d_chunks, nchunks, d_geo, c->lut_w, c->lut_h, TX, NT, d_segcnt, *dict, (uint16_t*)ws.rec16.p);
            } else {
            if ((rc = sl_optin(c, 13, (const void*)sl_count_lds_kernel<16, true>, 159 * 1024))) return rc;
            sl_count_lds_kernel<16, true><<<g, 64 * kCountWaves, lds_c, M>>>((const eorb_raw_event*)d_events, d_chunks, nchunks, d_geo, c->lut_w, c->lut_h, TX, NT, d_segcnt, *dict);
            }
        }
        else if (nchunks && cl_env && TX <= 127 && TY <= 127 && lds_c <= 159 * 1024) {
            const uint16_t* d_geo = (const uint16_t*)((const char*)c->sl_tab.p + sizeof(uint2) * nsrc);
            const int g = std::min(ncu, (nchunks + kCountWaves - 1) / kCountWaves);
            // 16-byte records on a sensor of at most 65 535 pixels: the pass leaves a 2-byte record per event for the scatter
            static const int tc_env = [] { const char* e = getenv("EORB_SLOT_TRANSCODE"); return e ? atoi(e) : 1; }();
            uint16_t* d_rec16 = nullptr;
            // ... or, where the rank scatter would run: an 8-byte record that also carries the ranks of the event's entries in the
            // chunk's runs (the values the counting atomics return), so that the scatter neither counts nor ranks (sl_scatter_pre_kernel)
            static const int pre_env = [] { const char* e = getenv("EORB_SLOT_PRERANK"); return e ? atoi(e) : 1; }();
            const int pre_on = c->dbg_slot_prerank >= 0 ? c->dbg_slot_prerank : pre_env;
            if (pre_on && (stride == 16 || stride == 4 || stride == 2) && nsrc <= 65535 && sc.rank && sl_lds_pre(NT, chunk) <= 158 * 1024) {
                if ((rc = ensure(c, ws.rec16, sizeof(uint64_t) * (size_t)std::max<int64_t>(h_offsets[B], 1)))) return rc;
                d_rec16 = (uint16_t*)ws.rec16.p;
                prerank = true;
#define SL_COUNTR(ST, BIT) do { if ((rc = sl_optin(c, BIT, (const void*)sl_count_lds_kernel<ST, false, true>, 159 * 1024))) return rc; \
                sl_count_lds_kernel<ST, false, true><<<g, 64 * kCountWaves, lds_c, M>>>(d_ev, d_chunks, nchunks, d_geo, c->lut_w, c->lut_h, TX, NT, d_segcnt, SlotDict{nullptr, 0u, nullptr, nullptr, 0}, d_rec16); } while (0)
                if (stride == 16) SL_COUNTR(16, 18); else if (stride == 4) SL_COUNTR(4, 19); else SL_COUNTR(2, 20);
#undef SL_COUNTR
            }
            else {
            if (tc_env && stride == 16 && nsrc <= 65535 && sc.rank) {
                if ((rc = ensure(c, ws.rec16, sizeof(uint16_t) * (size_t)std::max<int64_t>(h_offsets[B], 1)))) return rc;
                d_rec16 = (uint16_t*)ws.rec16.p;
            }
#define SL_COUNT(ST, BIT) do { if ((rc = sl_optin(c, BIT, (const void*)sl_count_lds_kernel<ST>, 159 * 1024))) return rc; \
                sl_count_lds_kernel<ST><<<g, 64 * kCountWaves, lds_c, M>>>(d_ev, d_chunks, nchunks, d_geo, c->lut_w, c->lut_h, TX, NT, d_segcnt, SlotDict{nullptr, 0u, nullptr, nullptr}, d_rec16); } while (0)
            if (stride == 16) SL_COUNT(16, 0); else if (stride == 4) SL_COUNT(4, 1); else if (stride == 2) SL_COUNT(2, 10); else SL_COUNT(-4, 2);
#undef SL_COUNT
            if (d_rec16) { d_ev = (const eorb_raw_event*)d_rec16; scat_stride = 2; }
            }
        }
        else if (nchunks) {
            if (stride == 16) sl_count_kernel<16><<<nchunks, 256, lds, M>>>(d_ev, d_chunks, d_tab, c->lut_w, c->lut_h, TX, NT, d_segcnt);
            else if (stride == 4) sl_count_kernel<4><<<nchunks, 256, lds, M>>>(d_ev, d_chunks, d_tab, c->lut_w, c->lut_h, TX, NT, d_segcnt);
            else if (stride == 2) sl_count_kernel<2><<<nchunks, 256, lds, M>>>(d_ev, d_chunks, d_tab, c->lut_w, c->lut_h, TX, NT, d_segcnt);
            else sl_count_kernel<-4><<<nchunks, 256, lds, M>>>(d_ev, d_chunks, d_tab, c->lut_w, c->lut_h, TX, NT, d_segcnt);
        }
        ps.reset(); ps.reset(new ProfScope(c, "ev_scan"));
        sl_scan_kernel<<<B, 1024, 0, M>>>(d_slice_c0, d_segcnt, NT, d_segbase, d_tile_cnt, d_tile_base);
        ps.reset();
        // ---- the gather's plan needs the scan's counts only: it runs on its own stream BESIDE the scatter (50 us of single-block
        //      kernels off the critical path) ----
        EORB_HIP(c, hipEventRecord(E[0], M));
        EORB_HIP(c, hipStreamWaitEvent(P, E[0], 0));
        EORB_HIP(c, hipMemsetAsync(ws.hot.p, 0, 256, P));
        sl_plan_kernel<<<NT, 256, sizeof(uint32_t) * 2 * (size_t)B, P>>>(d_tile_cnt, d_tile_base, d_slice_eb, B, NT, TX, hot_min, d_nslots, d_rowbase,
                                                                               d_items, d_tile_w, d_tile_m, d_ctr, d_hot_cnt, d_hot_items, hot_cap);
        sl_tasks_kernel<<<1, 1024, sizeof(uint32_t) * (size_t)NT, P>>>(d_tile_w, d_tile_m, NT, Gt, max_per_tile, d_scr, d_task, d_hot_cnt, hot_cap);
        EORB_HIP(c, hipEventRecord(E[1], P));
        // ---- the scatter (form and chunk size chosen up front: sl_choose_scatter) ----
        ProfScope ps2(c, "ev_scatter");
        const int stride = scat_stride;                                   // (what the count pass left for the scatter)
        if (nchunks && prerank) {
            const size_t lds_p = sl_lds_pre(NT, chunk);
            const uint16_t* d_segcnt_c = d_segcnt;
            if (sc.waves == 16) {
                if ((rc = sl_optin(c, 16, (const void*)sl_scatter_pre_kernel<16>, 159 * 1024))) return rc;
                sl_scatter_pre_kernel<16><<<nchunks, 64 * 16, lds_p, M>>>((const uint64_t*)ws.rec16.p, d_chunks, d_tab, c->lut_w * c->lut_h, TX, NT, chunk, d_slice_eb, d_segcnt_c, d_segbase, d_tile_base, (slot_entry*)ws.entries.p);
            } else {
                if ((rc = sl_optin(c, 17, (const void*)sl_scatter_pre_kernel<8>, 159 * 1024))) return rc;
                sl_scatter_pre_kernel<8><<<nchunks, 64 * 8, lds_p, M>>>((const uint64_t*)ws.rec16.p, d_chunks, d_tab, c->lut_w * c->lut_h, TX, NT, chunk, d_slice_eb, d_segcnt_c, d_segbase, d_tile_base, (slot_entry*)ws.entries.p);
            }
        }
        else if (nchunks && sc.rank) {
#define SL_SCAT(ST, NW, BIT) do { if ((rc = sl_optin(c, BIT, (const void*)sl_scatter_rank_kernel<ST, NW>, 159 * 1024))) return rc; \
                sl_scatter_rank_kernel<ST, NW><<<nchunks, 64 * NW, sc.lds, M>>>(d_ev, d_chunks, d_tab, c->lut_w, c->lut_h, TX, NT, chunk, \
                                                                                   d_slice_eb, d_segbase, d_tile_base, (slot_entry*)ws.entries.p); } while (0)
            if (sc.waves == 16) { if (stride == 16) SL_SCAT(16, 16, 3); else if (stride == 4) SL_SCAT(4, 16, 4); else if (stride == 2) SL_SCAT(2, 16, 11); else SL_SCAT(-4, 16, 5); }
            else { if (stride == 16) SL_SCAT(16, 8, 6); else if (stride == 4) SL_SCAT(4, 8, 7); else if (stride == 2) SL_SCAT(2, 8, 12); else SL_SCAT(-4, 8, 8); }
#undef SL_SCAT
        }
        else if (nchunks)
            sl_scatter_kernel<<<nchunks, 64 * kSlotScatWaves, sc.lds, M>>>(d_ev, d_chunks, d_tab, stride, c->lut_w, c->lut_h, TX, TY, NT, chunk,
                                                                                 d_slice_eb, d_segbase, d_tile_base, (slot_entry*)ws.entries.p);
        EORB_LAUNCH_CHECK(c, "ev_bin (slot) kernels");
    }
    EORB_HIP(c, hipEventRecord(E[2], M));                                // the part's entries are in place
    {
        SlotGather Pg{d_task, d_items, (const uint8_t*)ws.entries.p, d_nslots, d_rowbase, (const float*)c->sl_rows.p,
                      d_tile_w, d_ctr, d_f32, d_minmax_enc, d_info, (int*)c->status.p, B, W, H, TX, NT, c->sl_null, prio_ref, nullptr};
#ifdef EORB_SLOT_TRACE
        if ((rc = ensure(c, c->sl_trace, sizeof(unsigned long long) * 6 * 16 * (size_t)Gt + 64))) return rc;
        EORB_HIP(c, hipMemsetAsync(c->sl_trace.p, 0, sizeof(unsigned long long) * 6 * 16 * (size_t)Gt + 64, M));
        Pg.trace = (unsigned long long*)c->sl_trace.p + 8;
        c->sl_trace_n = (long long)Gt * 16;
#endif
        if ((rc = sl_optin(c, 9, (const void*)sl_gather_kernel, 160 * 1024))) return rc;
        if (hot_min) {
            // the long lists on the high-priority stream beside the gather: they need the scatter's entries and the plan's descriptors
            static const int hw_env = [] { const char* e = getenv("EORB_SLOT_HOT_WAVES"); return e ? atoi(e) : 0; }();
            const int hw = c->dbg_slot_hot_waves > 0 ? c->dbg_slot_hot_waves : (hw_env > 0 ? hw_env : 8 * ncu);     // two per SIMD: all of its registers
            EORB_HIP(c, hipStreamWaitEvent(Hs, E[2], 0));
            EORB_HIP(c, hipStreamWaitEvent(Hs, E[1], 0));
            sl_hot_kernel<<<hw, 64, 0, Hs>>>(d_hot_cnt, d_hot_cnt + 32, d_hot_items, kHotCap, (const float*)c->sl_rows.p, (const slot_entry*)ws.entries.p,
                                             d_f32, d_minmax_enc, W, H);
        }
        EORB_HIP(c, hipEventRecord(E[3], Hs));
        if (G != M) EORB_HIP(c, hipStreamWaitEvent(G, E[2], 0));
        EORB_HIP(c, hipStreamWaitEvent(G, E[1], 0));                       // the plan and the task table
        {
            ProfScope ps(c, "ev_gather", G);
            sl_gather_kernel<<<Gt, 64 * nw, lds_g, G>>>(Pg);
        }
        EORB_HIP(c, hipEventRecord(E[4], G));
        EORB_LAUNCH_CHECK(c, "sl_gather_kernel");
    }
    return EORB_OK;
}

// The slot form for B slices of raw events (no polarity, Gaussian stamp).  On request a batch runs as two halves: the LDS gather and
// the register-row kernel of the first half (vector-ALU / LDS-read / scalar-ALU bound) beside the count and scatter passes of the
// second (HBM- and LDS-atomic bound).  Returns kSlotDeclined (> 0, nothing launched) when the batch's shape does not
// fit: the caller goes on with the batch pipeline.
int ev_slots_accumulate(eorb_ctx* c, const void* d_events, int stride, const int64_t* h_offsets, int B, int W, int H, int TX, int TY,
                        float* d_f32, uint32_t* d_minmax_enc, const SlotDict* dict)
{
    const int NT = TX * TY;
    const int64_t nev = h_offsets[B] - h_offsets[0];
    static const int rank_env0 = [] { const char* e = getenv("EORB_SLOT_RANK"); return e ? atoi(e) : 1; }();
    const bool rank_allowed = c->sl_rank_ok == 1 && (c->dbg_slot_rank < 0 ? rank_env0 != 0 : c->dbg_slot_rank != 0);
    static const int halves_env = [] { const char* e = getenv("EORB_SLOT_HALVES"); return e ? atoi(e) : -1; }();
    const int halves_opt = c->dbg_slot_halves >= 0 ? c->dbg_slot_halves : halves_env;
    // Off by default.  Measured at 128 x 1 Mev: 3.94 ms per step in halves against 3.34 in one part -- each half's gather phase is bounded
    // from below by the serial chain of its own longest list (200 000 entries x 13 cycles = 1.1 ms on the register-row kernel's
    // wavefront), so two halves pay that chain twice; the overlap of the second half's binning with the first half's gather does not
    // buy it back.  The parts machinery stays (test hook "slot_halves", EORB_SLOT_HALVES=1) for batches without such chains.
    const int nparts = (halves_opt > 0 && B >= 2) ? 2 : 1;
    const int64_t per_slice = nev / B;
    SlotScatterChoice sc;
    // sl_plan_kernel keeps two words per slice in LDS and ranks a position's lists by an O(B^2) loop, sl_scan_kernel is one workgroup
    // per slice: batches of more than 2 048 slices take the batch pipeline; so do tile grids whose count rows do not fit the LDS
    if (B > 2048 || (size_t)NT * 4 > 64 * 1024 || !sl_choose_scatter(c, NT, per_slice, rank_allowed, &sc)) return kSlotDeclined;
    for (int b = 0; b < B; b++) {
        const int64_t s = h_offsets[b], e = h_offsets[b + 1];
        if (e < s) return set_err(c, EORB_E_ARG, "ev_accumulate: offsets not monotone");
        if ((e - s) * 4 + (int64_t)NT * 16 >= (int64_t)1 << 31) return set_err(c, EORB_E_CAPACITY, "ev_accumulate: %lld events in one slice", (long long)(e - s));
    }
    int rc;
    if ((rc = sl_streams(c))) return rc;
    c->sl_calls++;
    c->sl_last_rank = sc.rank ? 1 : 0; c->sl_last_chunk = sc.chunk; c->sl_last_parts = nparts;
    const int B0 = nparts == 2 ? B / 2 : B;
    for (int part = 0; part < nparts; part++) {
        const int b0 = part ? B0 : 0, nb_ = part ? B - B0 : B0;
        if ((rc = slots_part(c, c->sl_ws[part], part, nparts, d_events, stride, h_offsets + b0, nb_, W, H, TX, TY,
                             d_f32 + (size_t)b0 * W * H, d_minmax_enc + 2 * (size_t)b0, sc, dict))) return rc;
    }
    // whatever the other streams did is done before the images are read (streams are in order: the last part's events cover the first's)
    hipEvent_t* E = c->sl_ev + 5 * (nparts - 1);
    EORB_HIP(c, hipStreamWaitEvent(c->stream, E[3], 0));
    if (nparts > 1) EORB_HIP(c, hipStreamWaitEvent(c->stream, E[4], 0));
    return EORB_OK;
}

// EORB_SLOT_TRACE builds: the per-wave records of the last gather (tools/slot_trace.py)
int ev_slots_trace_read(eorb_ctx* c, unsigned long long* out, long long max_records)
{
    const long long n = std::min<long long>(c->sl_trace_n, max_records);
    if (n <= 0 || !c->sl_trace.p) return 0;
    if (hipMemcpy(out, (unsigned long long*)c->sl_trace.p + 8, sizeof(unsigned long long) * 6 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)n;
}

}  // namespace eorb
