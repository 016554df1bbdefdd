// ev_accum.hip -- event -> image accumulation on gfx950, bit-exact w.r.t. the sequential CPU loop.
//
// Replaces EvImConverter::ev2im_gauss / ev2im (src/Event/EventConversion.cc:173-269 of the
// reference).  The reference adds one (2h+1)^2 Gaussian stamp per event into a float image
// SEQUENTIALLY; float addition is not associative, so every pixel must receive its contributions in
// event order.  Design (SURVEY App.B H1):
//   K1 ev_bin_kernel     one wavefront per chunk of kChunk consecutive events: stable binning of the
//                        chunk's events into 8x8-pixel tiles (an event is copied to every tile its stamp
//                        touches).  Counts with LDS atomics, wave scan, then an ORDER-PRESERVING scatter:
//                        lanes = consecutive events, rank among same-tile lanes by ballot matching.
//                        Tiles are visited in parity classes so a tile is only ever targeted in one pass.
//   K2 ev_gather_kernel  one wavefront per tile (lane = pixel): walks the tile's segments chunk by chunk
//                        (= event order), evaluates the one stamp tap that hits its pixel and adds it.
//                        Running min/max (resolveMinMaxVals :32-39) reduced per wave -> atomics.
//   K3 ev_normalize_kernel  normalizeImage (:67-72): convertTo(CV_8UC1, alpha, beta), round-half-even.
// All kernels are batched over time-slices (blockIdx ranges over slices x tiles / chunks).
#include "eorb_ctx.h"
#include "dev_math.h"
#include <math.h>
#include <algorithm>
#include <string.h>
#include <vector>

namespace eorb {

__device__ __forceinline__ uint32_t enc_f32(float f)
{   // order-preserving map float -> uint32 (for atomicMax/atomicMin on floats of either sign)
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e)
{
    uint32_t u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    return __uint_as_float(u);
}

struct BinParams {
    int W, H, h;          // image size, stamp half window (0 in count mode)
    int TX, TY, NT;       // tiles
    int nbits;            // bits needed for a tile id
    int cap;              // entries per chunk
    int mode_count;       // 1: ev2im (coords rounded, no stamp)
    int pol;
};

template <int R, bool POL>
__global__ __launch_bounds__(64) void ev_bin_kernel(const eorb_event16* __restrict__ ev,
                                                    const ChunkDesc* __restrict__ chunks, BinParams P,
                                                    uint16_t* __restrict__ segoff, float* __restrict__ entries)
{
    extern __shared__ uint32_t cnt[];               // NT + 1
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    const ChunkDesc cd = chunks[chunk];
    const int NT = P.NT;
    for (int i = lane; i <= NT; i += 64) cnt[i] = 0;
    __syncthreads();
    const eorb_event16* e = ev + cd.start;
    constexpr int ESZ = POL ? 4 : 2;

    // ---- pass A: count ----
    for (int s = 0; s < cd.n; s += 64) {
        const int k = s + lane;
        if (k < cd.n) {
            const float x = e[k].x, y = e[k].y;
            const int xi = P.mode_count ? (int)roundf(x) : (int)floorf(x);
            const int yi = P.mode_count ? (int)roundf(y) : (int)floorf(y);
            int tx0 = (xi - P.h) >> 3, tx1 = (xi + P.h) >> 3, ty0 = (yi - P.h) >> 3, ty1 = (yi + P.h) >> 3;
            tx0 = max(tx0, 0); ty0 = max(ty0, 0); tx1 = min(tx1, P.TX - 1); ty1 = min(ty1, P.TY - 1);
            for (int ty = ty0; ty <= ty1; ty++)
                for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&cnt[ty * P.TX + tx], 1u);
        }
    }
    __syncthreads();
    // ---- exclusive scan -> segment offsets ----
    uint32_t running = 0;
    uint16_t* so = segoff + (size_t)chunk * (NT + 1);
    for (int i0 = 0; i0 < NT; i0 += 64) {
        const int i = i0 + lane;
        uint32_t v = (i < NT) ? cnt[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        const uint32_t excl = running + incl - v;
        if (i < NT) { cnt[i] = excl; so[i] = (uint16_t)excl; }
        running += __shfl(incl, 63, 64);
    }
    if (lane == 0) so[NT] = (uint16_t)running;
    __syncthreads();
    // ---- pass B: order-preserving scatter ----
    float* out = entries + (size_t)chunk * P.cap * ESZ;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int s = 0; s < cd.n; s += 64) {
        const int k = s + lane;
        const bool valid = k < cd.n;
        float x = 0.f, y = 0.f, sg = 1.f;
        int tx0 = 1, tx1 = 0, ty0 = 1, ty1 = 0;
        if (valid) {
            x = e[k].x; y = e[k].y;
            if (POL) sg = (__double_as_longlong(e[k].t) < 0) ? -1.0f : 1.0f;
            const int xi = P.mode_count ? (int)roundf(x) : (int)floorf(x);
            const int yi = P.mode_count ? (int)roundf(y) : (int)floorf(y);
            tx0 = max((xi - P.h) >> 3, 0); tx1 = min((xi + P.h) >> 3, P.TX - 1);
            ty0 = max((yi - P.h) >> 3, 0); ty1 = min((yi + P.h) >> 3, P.TY - 1);
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
                    const uint32_t pos = base + rank;
                    if (POL) { float4 v = make_float4(x, y, sg, 0.f); *(float4*)(out + (size_t)pos * 4) = v; }
                    else { float2 v = make_float2(x, y); *(float2*)(out + (size_t)pos * 2) = v; }
                    if (rank == 0) cnt[key] = base + (uint32_t)__popcll(m);
                }
                __syncthreads();
            }
        }
    }
}

struct GatherParams {
    int W, H, h, TX, TY, NT, cap;
    int mode_count;
    int total;           // number of (slice, tile) work items
    float two_sig2;      // 2.0f * sig2
    float norm;          // 2.0f*float(CV_PI)*sig2
};

template <bool POL>
__global__ __launch_bounds__(64) void ev_gather_kernel(const int* __restrict__ slice_chunk0,   // B+1
                                                       GatherParams P, const uint16_t* __restrict__ segoff,
                                                       const float* __restrict__ entries,
                                                       float* __restrict__ img, uint32_t* __restrict__ minmax_enc)
{
    __shared__ uint64_t tab[32];
    const int lane = threadIdx.x;
    if (lane < 32) tab[lane] = kExp2Tab[lane];
    __syncthreads();
    // XCD-aware mapping: blocks b and b+8 share an XCD/L2; give each XCD a contiguous run of tiles so
    // neighbouring tiles (adjacent segments of the same chunks) hit the same L2.
    const int nb = gridDim.x;
    const int per = (nb + 7) / 8;
    int logical = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (logical >= P.total) return;     // grid is padded to a multiple of 8 by the launcher
    const int slice = logical / P.NT;
    const int tile = logical - slice * P.NT;
    const int c0 = slice_chunk0[slice], c1 = slice_chunk0[slice + 1];
    const int px = (tile % P.TX) * kTile + (lane & 7);
    const int py = (tile / P.TX) * kTile + (lane >> 3);
    const bool inimg = px < P.W && py < P.H;
    constexpr int ESZ = POL ? 4 : 2;
    float acc = 0.0f, vmax = -1000000.0f, vmin = 0.0f;
    for (int c = c0; c < c1; c++) {
        const uint16_t* so = segoff + (size_t)c * (P.NT + 1) + tile;
        const int o0 = so[0], o1 = so[1];
        const float* base = entries + (size_t)c * P.cap * ESZ;
        for (int j0 = o0; j0 < o1; j0 += 64) {
            const int j = j0 + lane;
            int exi = 0, eyi = 0; float exr = 0.f, eyr = 0.f, esg = 1.f;
            if (j < o1) {
                float ex, ey;
                if (POL) { float4 v = *(const float4*)(base + (size_t)j * 4); ex = v.x; ey = v.y; esg = v.z; }
                else { float2 v = *(const float2*)(base + (size_t)j * 2); ex = v.x; ey = v.y; }
                if (P.mode_count) {                    // roundFloatCoord (:46-49)
                    exi = (int)roundf(ex); eyi = (int)roundf(ey);
                } else {                               // breakFloatCoords (:51-57)
                    const float fx = floorf(ex), fy = floorf(ey);
                    exi = (int)fx; eyi = (int)fy;
                    exr = ex - (float)exi; eyr = ey - (float)eyi;
                }
            }
            const int cnt = min(64, o1 - j0);
            for (int q = 0; q < cnt; q++) {
                const int xi = __builtin_amdgcn_readlane(exi, q);
                const int yi = __builtin_amdgcn_readlane(eyi, q);
                const int dx = px - xi, dy = py - yi;
                if ((unsigned)(dx + P.h) <= (unsigned)(2 * P.h) && (unsigned)(dy + P.h) <= (unsigned)(2 * P.h)) {
                    float val;
                    if (P.mode_count) {
                        val = 0.001f;
                    } else {
                        const float xr = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(exr), q));
                        const float yr = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(eyr), q));
                        const float fx = (float)dx - xr, fy = (float)dy - yr;      // exp_XY2f(i-xRes, j-yRes) :59-65
                        const float xx = fx * fx, yy = fy * fy;
                        float dd = xx + yy;
                        dd = dd / P.two_sig2;
                        val = dev_expf_nonpos(-dd, tab) / P.norm;
                    }
                    float sgn = 1.0f;
                    if (POL) sgn = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(esg), q));
                    const float nv = acc + sgn * val;
                    acc = nv;
                    vmax = fmaxf(vmax, nv);
                    vmin = fminf(vmin, nv);
                }
            }
        }
    }
    if (inimg) img[(size_t)slice * P.W * P.H + (size_t)py * P.W + px] = acc;
    else { vmax = -1000000.0f; vmin = 0.0f; }
    // wave reduce min/max
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
int ev_accumulate_dev(eorb_ctx* c, const eorb_event16* d_ev, const int64_t* h_offsets, int B, int W, int H,
                      float sigma, int pol, int mode_count, float* d_f32, uint8_t* d_u8, int normalized,
                      uint32_t* d_minmax_enc)
{
    if (B <= 0 || W <= 0 || H <= 0) return set_err(c, EORB_E_ARG, "ev_accumulate: bad size");
    const int h = mode_count ? 0 : (int)ceil((double)sigma * 3.0);     // lenHalfWin :222
    if (h > 8) return set_err(c, EORB_E_CONFIG, "ev_accumulate: sigma %.3f gives half window %d > 8", sigma, h);
    const int R = (2 * h <= kTile) ? ((h == 0) ? 1 : 2) : 3;            // max tiles an event spans per axis
    const int TX = (W + kTile - 1) / kTile, TY = (H + kTile - 1) / kTile, NT = TX * TY;
    int nbits = 1; while ((1 << nbits) < NT) nbits++;
    const int dup = R * R;
    const int cap = kChunk * dup;
    if (cap > 65535) return set_err(c, EORB_E_CONFIG, "ev_accumulate: chunk capacity overflow");
    // chunk list (host) -> device
    std::vector<ChunkDesc> cds;
    std::vector<int> slice_c0(B + 1);
    for (int b = 0; b < B; b++) {
        slice_c0[b] = (int)cds.size();
        const int64_t s = h_offsets[b], e = h_offsets[b + 1];
        if (e < s) return set_err(c, EORB_E_ARG, "ev_accumulate: offsets not monotone");
        for (int64_t k = s; k < e; k += kChunk) {
            ChunkDesc cd; cd.start = k; cd.n = (int32_t)std::min<int64_t>(kChunk, e - k); cd.slice = b;
            cds.push_back(cd);
        }
    }
    slice_c0[B] = (int)cds.size();
    const int nchunks = (int)cds.size();
    const size_t cd_bytes = sizeof(ChunkDesc) * (size_t)std::max(nchunks, 1);
    const size_t sc_bytes = sizeof(int) * (size_t)(B + 1);
    int rc;
    if ((rc = ensure(c, c->chunks, cd_bytes + sc_bytes))) return rc;
    if ((rc = ensure(c, c->segoff, sizeof(uint16_t) * (size_t)std::max(nchunks, 1) * (NT + 1)))) return rc;
    const int esz = pol ? 4 : 2;
    if ((rc = ensure(c, c->entries, sizeof(float) * esz * (size_t)std::max(nchunks, 1) * cap))) return rc;
    char* hp = (char*)pinned(c, cd_bytes + sc_bytes);
    if (!hp) return set_err(c, EORB_E_HIP, "pinned alloc failed");
    if (nchunks) memcpy(hp, cds.data(), sizeof(ChunkDesc) * nchunks);
    memcpy(hp + cd_bytes, slice_c0.data(), sc_bytes);
    EORB_HIP(c, hipMemcpyAsync(c->chunks.p, hp, cd_bytes + sc_bytes, hipMemcpyHostToDevice, c->stream));
    const ChunkDesc* d_chunks = (const ChunkDesc*)c->chunks.p;
    const int* d_slice_c0 = (const int*)((char*)c->chunks.p + cd_bytes);

    {
        ProfScope ps(c, "ev_minmax_init");
        ev_minmax_init_kernel<<<(B + 63) / 64, 64, 0, c->stream>>>(d_minmax_enc, B);
    }
    if (nchunks) {
        BinParams P{W, H, h, TX, TY, NT, nbits, cap, mode_count, pol};
        const size_t lds = sizeof(uint32_t) * (NT + 1);
        ProfScope ps(c, "ev_bin");
        uint16_t* so = (uint16_t*)c->segoff.p; float* en = (float*)c->entries.p;
#define LAUNCH_BIN(RR, PP) ev_bin_kernel<RR, PP><<<nchunks, 64, lds, c->stream>>>(d_ev, d_chunks, P, so, en)
        if (R == 1) { if (pol) LAUNCH_BIN(1, true); else LAUNCH_BIN(1, false); }
        else if (R == 2) { if (pol) LAUNCH_BIN(2, true); else LAUNCH_BIN(2, false); }
        else { if (pol) LAUNCH_BIN(3, true); else LAUNCH_BIN(3, false); }
#undef LAUNCH_BIN
        EORB_LAUNCH_CHECK(c, "ev_bin_kernel");
    }
    {
        const float sig2 = sigma * sigma;
        GatherParams G{W, H, h, TX, TY, NT, cap, mode_count, B * NT, 2.0f * sig2,
                       2.0f * (float)3.1415926535897932384626433832795 * sig2};
        const int nb = B * NT;
        const int grid = ((nb + 7) / 8) * 8;
        // the kernel derives `per` from gridDim; pass the padded grid and let surplus blocks exit
        GatherParams G2 = G;
        ProfScope ps(c, "ev_gather");
        if (pol) ev_gather_kernel<true><<<grid, 64, 0, c->stream>>>(d_slice_c0, G2, (const uint16_t*)c->segoff.p,
                                                                   (const float*)c->entries.p, d_f32, d_minmax_enc);
        else ev_gather_kernel<false><<<grid, 64, 0, c->stream>>>(d_slice_c0, G2, (const uint16_t*)c->segoff.p,
                                                                 (const float*)c->entries.p, d_f32, d_minmax_enc);
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

int ev_decode_minmax(eorb_ctx* c, const uint32_t* d_enc, float* d_out, int B)
{
    ev_decode_minmax_kernel<<<(2 * B + 63) / 64, 64, 0, c->stream>>>(d_enc, d_out, B);
    EORB_LAUNCH_CHECK(c, "ev_decode_minmax_kernel");
    return EORB_OK;
}

}  // namespace eorb
