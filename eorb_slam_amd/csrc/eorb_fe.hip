// eorb_fe.hip -- C ABI of libeorb_fe.so (include/eorb_fe.h): context, host-buffer entry points and the
// batched HBM-resident front end.  No CPU fallback anywhere: every entry point launches HIP kernels.
#include "eorb_ctx.h"
#include "match_args.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace eorb {

int ev_decode_minmax(eorb_ctx* c, const uint32_t* d_enc, float* d_out, int B);
int ev_divcheck(eorb_ctx* c, float lo, float hi, float sigma, unsigned long long* bad_out);
int ev_diag_read(unsigned long long* out16);
int ev_trace_read(unsigned long long* out, int n);
int ev_warp_se3_dev(eorb_ctx* c, const eorb_event16* d_in, eorb_event16* d_out, int n, const eorb_camera* cam, double angle,
                    const double axis[3], const double tt[3], float medDepth, const float* d_depth);
int ev_warp_se2_dev(eorb_ctx* c, const eorb_event16* d_in, eorb_event16* d_out, int n, const eorb_camera* cam, const float* params, int nparams);
int ev_focus_dev(eorb_ctx* c, const float* d_img, int nimg, int W, int H, float* d_out);
int ev_cvnormalize_dev(eorb_ctx* c, const float* d_img, int npix, uint32_t* d_mm, uint8_t* d_out);
int ev_mathhash(eorb_ctx* c, int which, uint32_t lo_bits, uint32_t hi_bits, unsigned long long* out);
int ev_cvnormalize_n_dev(eorb_ctx* c, const float* d_imgs, int nimg, int npix, uint32_t* d_mm, uint8_t* d_outs);
int ev_contest_select_dev(eorb_ctx* c, const float* d_focus_img, const int img_of[4], int half_img, const uint8_t* d_u8s, int npix,
                          float* d_focus_out, int* d_winner, uint8_t* d_out);
int ev_kp_points_dev(eorb_ctx* c, const eorb_keypoint* d_kps, const int32_t* d_n, int cap, float* d_pts);
int orb_configure(eorb_ctx* c, const eorb_orb_params* p, int W, int H);
int orb_err_flag(eorb_ctx* c, int B, int* flag);
int orb_err_flag_to(eorb_ctx* c, int B, int32_t* d_dst);
int bf_knn2_dev(eorb_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx2, int32_t* d_dist2);
int search_proj_last_dev(eorb_ctx* c, const eorb_keypoint* cur_kps, int n_cur, const uint8_t* cur_desc, int cur_stride,
                         const uint8_t* cur_is_orb, const eorb_keypoint* last_kps, int n_last, const uint8_t* last_is_orb,
                         const uint8_t* valid, const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
                         int dist_th, eorb_grid_bounds gb, int32_t* cur_mp, float th, int mode, int checkOri,
                         int32_t* nmatches, const float* cur_uright = nullptr, const float* q_ur = nullptr);
int search_proj_map_dev(eorb_ctx* c, const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const uint8_t* is_orb,
                        int M, const uint8_t* in_view, const float4* mp_f4 /* projX, projY, viewCos, levelScale */,
                        const int32_t* level, const uint8_t* mp_desc, const uint8_t* mp_obs, const uint8_t* mp_is_orb,
                        eorb_grid_bounds gb, int32_t* frame_mp, float th, float nnratio, int32_t* nmatches,
                        const float* uright = nullptr, const float* q_ur = nullptr);

int orb_pyramid_blur_dev(eorb_ctx* c, const uint8_t* d_img, int img_stride);
int orb_tracked_dev(eorb_ctx* c, eorb_keypoint* d_kps, int n, int mode, const uint8_t* d_ref, uint8_t* d_desc, uint8_t* d_oob);
int search_bow_dev(eorb_ctx* c, const eorb_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_has_mp,
                   const uint32_t* kf_nodes, const int32_t* kf_off, const int32_t* kf_idx, int kf_nn,
                   const eorb_keypoint* f_kps, int n_f, const uint8_t* f_desc, const uint32_t* f_nodes, const int32_t* f_off,
                   const int32_t* f_idx, int f_nn, int32_t* match_f, int8_t* bin_f, int32_t* histo, int32_t* nmatches,
                   float nnratio, int checkOri, int kf_kf, const uint8_t* f_has_mp, int32_t* match12, int n_kf);
int search_tri_dev(eorb_ctx* c, const TriArgs& A);
int kf_radius_dev(eorb_ctx* c, const RadArgs& A, uint16_t* d_cell);
int bow_transform_dev(eorb_ctx* c, const uint8_t* d_desc, int n, int stride, const BowVoc& V, int levelsup, int weighting, int norm,
                      uint32_t* d_word_of, double* d_w_of, uint32_t* d_node_of, uint32_t* d_bow_word, double* d_bow_val,
                      uint32_t* d_fv_node, int32_t* d_fv_off, int32_t* d_fv_idx, int32_t* d_counts);
int window_match_dev(eorb_ctx* c, const uint8_t* d_q, int nq, int q_stride, const uint8_t* d_t, int t_stride, const int32_t* d_off,
                     const int32_t* d_cand, int32_t* d_out);
int distinctive_dev(eorb_ctx* c, const uint8_t* d_desc, const int32_t* d_offsets, int M, int32_t* d_best);
int sort_response_dev(eorb_ctx* c, const eorb_keypoint* d_kps, int n, int32_t* d_perm);

int set_err(eorb_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}

int hip_check(eorb_ctx* c, hipError_t e, const char* what)
{
    return set_err(c, EORB_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

int ensure(eorb_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return EORB_OK;
    if (b.p) { hipError_t e = hipFree(b.p); b.p = nullptr; b.cap = 0; if (e != hipSuccess) return hip_check(c, e, "hipFree"); }
    const size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return hip_check(c, e, "hipMalloc"); }
    b.cap = want;
    return EORB_OK;
}

// ---- buffers that travel one by one (up() / down()): pinned bump allocators + copy kernels ----
// A pageable hipMemcpyAsync is staged by the runtime and waits; eight of them plus three downloads made a 19 us matcher a 110 us call.
// up(): memcpy into pinned memory, a kernel reads it over the link; down(): a kernel writes pinned memory, the host copy happens at the
// entry's stream wait (fe_stream_sync), which also empties both allocators.  What does not fit without waiting goes through the copy engine.
// up to eight copies in one launch (blockIdx.y = segment): an entry's buffers go up together and its results come back together
struct CopySegs8 { unsigned char* dst[8]; const unsigned char* src[8]; size_t n[8]; int align[8]; int count; };
__global__ void copy_segs_kernel(CopySegs8 S)
{
    const int sg = blockIdx.y;
    const unsigned char* src = S.src[sg]; unsigned char* dst = S.dst[sg];
    const size_t n = S.n[sg];
    const int align = S.align[sg];
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    if (align == 16) {
        const size_t n16 = n >> 4;
        for (size_t i = t; i < n16; i += nt) ((uint4*)dst)[i] = ((const uint4*)src)[i];
        for (size_t i = (n16 << 4) + t; i < n; i += nt) dst[i] = src[i];
    } else if (align == 4) {
        const size_t n4 = n >> 2;
        for (size_t i = t; i < n4; i += nt) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
        for (size_t i = (n4 << 2) + t; i < n; i += nt) dst[i] = src[i];
    } else
        for (size_t i = t; i < n; i += nt) dst[i] = src[i];
}
static int launch_copy_queue(eorb_ctx* c, std::vector<eorb_ctx::CopySeg>& q)
{
    for (size_t i0 = 0; i0 < q.size(); i0 += 8) {
        CopySegs8 S{};
        size_t units = 1;
        S.count = (int)std::min<size_t>(8, q.size() - i0);
        for (int k = 0; k < S.count; k++) {
            const auto& e = q[i0 + k];
            const uintptr_t a = (uintptr_t)e.src | (uintptr_t)e.dst;
            S.dst[k] = (unsigned char*)e.dst; S.src[k] = (const unsigned char*)e.src; S.n[k] = e.n;
            S.align[k] = !(a & 15) ? 16 : (!(a & 3) ? 4 : 1);
            units = std::max(units, (e.n + (size_t)S.align[k] - 1) / (size_t)S.align[k]);
        }
        copy_segs_kernel<<<dim3((unsigned)std::min<size_t>((units + 255) / 256, 128), (unsigned)S.count), 256, 0, c->stream>>>(S);
        EORB_LAUNCH_CHECK(c, "copy_segs_kernel");
    }
    q.clear();
    return EORB_OK;
}
// the uploads staged by up() so far, in one launch: call behind an entry's last up(), in front of its first kernel
int up_flush(eorb_ctx* c) { return c->up_queue.empty() ? EORB_OK : launch_copy_queue(c, c->up_queue); }
static void* pin_bump(eorb_ctx::PinBump& b, size_t bytes)            // 256-byte granules; nullptr: not without waiting for queued work
{
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (b.used + need > b.cap) {
        if (b.used) return nullptr;
        if (b.p) { hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
        const size_t want = std::max<size_t>(2 * need, (size_t)1 << 20);
        if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) { b.p = nullptr; return nullptr; }
        b.cap = want;
    }
    void* r = (char*)b.p + b.used;
    b.used += need;
    return r;
}
void pinned_release_lazy(eorb_ctx* c);
// every stream wait of an entry point: the queued copies are done -- results to their host destinations, the staging memory free again
hipError_t fe_stream_sync(eorb_ctx* c)
{
    if (!c->up_queue.empty() && launch_copy_queue(c, c->up_queue) != EORB_OK) return hipErrorUnknown;
    if (!c->dn_queue.empty() && launch_copy_queue(c, c->dn_queue) != EORB_OK) return hipErrorUnknown;
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) {
        for (const auto& d : c->dn_pending) memcpy(d.dst, (const char*)c->dn_pin.p + d.off, d.bytes);
        pinned_release_lazy(c);
    }
    c->dn_pending.clear(); c->dn_queue.clear(); c->up_queue.clear(); c->dn_pin.used = 0; c->up_pin.used = 0;
    return e;
}
// entry prologue: the context's device; whatever an entry that failed half-way left pending is dropped (its destinations may be gone)
void fe_enter(eorb_ctx* c)
{
    hipSetDevice(c->device);
    if (!c->dn_pending.empty()) c->dn_pending.clear();
    if (!c->dn_queue.empty()) c->dn_queue.clear();
    if (!c->up_queue.empty()) c->up_queue.clear();
}
static int down(eorb_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return EORB_OK;
    static const long kmax = [] { const char* e = getenv("EORB_DOWNLOAD_KERNEL_MAX"); return e ? atol(e) : (1L << 20); }();
    void* p = (long)bytes <= kmax ? pin_bump(c->dn_pin, bytes) : nullptr;
    if (!p) { EORB_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)); return EORB_OK; }
    c->dn_queue.push_back({p, src, bytes});
    c->dn_pending.push_back({dst, (size_t)((char*)p - (char*)c->dn_pin.p), bytes});
    return EORB_OK;
}

void* pinned(eorb_ctx* c, size_t bytes)
{
    eorb_ctx::PinnedSlot& s = c->pinned[c->pinned_next];
    c->pinned_cur = c->pinned_next;
    c->pinned_next = (c->pinned_next + 1) % eorb_ctx::kPinnedSlots;
    if (s.busy) {                                                   // the copy / kernel that read this slot has completed
        if (s.lazy) { hipStreamSynchronize(c->stream); pinned_release_lazy(c); }
        else hipEventSynchronize(s.ev);
        s.busy = false; s.lazy = false;
    }
    if (s.cap >= bytes) return s.p;
    if (s.p) { hipHostFree(s.p); s.p = nullptr; s.cap = 0; }
    const size_t want = bytes + bytes / 2 + 4096;
    if (hipHostMalloc(&s.p, want, hipHostMallocDefault) != hipSuccess) { s.p = nullptr; return nullptr; }
    s.cap = want;
    return s.p;
}

void pinned_release_lazy(eorb_ctx* c)
{
    for (auto& s : c->pinned) if (s.busy && s.lazy) { s.busy = false; s.lazy = false; }
}

// lazy: the host-buffer entry points wait for their results before they return, which covers every read of their staging slot: an event
// per slot was a marker in the queue in front of the call's next kernel (5 us of idle GPU per call in the time line)
void pinned_commit(eorb_ctx* c, bool lazy)
{
    if (c->pinned_cur < 0) return;
    eorb_ctx::PinnedSlot& s = c->pinned[c->pinned_cur];
    if (lazy) { s.busy = true; s.lazy = true; return; }
    if (!s.ev && hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) { s.ev = nullptr; hipStreamSynchronize(c->stream); return; }
    if (hipEventRecord(s.ev, c->stream) != hipSuccess) { hipStreamSynchronize(c->stream); return; }
    s.busy = true;
}

ProfScope::ProfScope(eorb_ctx* cc, const char* name, hipStream_t stream) : c(cc), idx(-1), st(stream ? stream : cc->stream)
{
    if (!c->prof) return;
    if (!c->prof_only.empty() && c->prof_only.find(std::string(",") + name + ",") == std::string::npos) return;
    for (size_t i = 0; i < c->profs.size(); i++) if (c->profs[i].name == name) { idx = (int)i; break; }
    if (idx < 0) { c->profs.emplace_back(); c->profs.back().name = name; idx = (int)c->profs.size() - 1; }
    // events come from a pool (creating a pair costs several microseconds: visible on the one-frame-per-call paths)
    auto take = [&](hipEvent_t& e) { if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else hipEventCreate(&e); };
    take(a); take(b);
    hipEventRecord(a, st);
}
ProfScope::~ProfScope()
{
    if (idx < 0) return;
    hipEventRecord(b, st);
    c->profs[idx].pending.emplace_back(a, b);
    c->profs[idx].launches++;
}

void prof_collect(eorb_ctx* c)
{
    for (auto& p : c->profs) {
        for (auto& ev : p.pending) {
            hipEventSynchronize(ev.second);
            float ms = 0; hipEventElapsedTime(&ms, ev.first, ev.second);
            p.total_ms += ms;
            c->ev_pool.push_back(ev.first); c->ev_pool.push_back(ev.second);
        }
        p.pending.clear();
    }
}

int* readback_buf(eorb_ctx* c)
{
    if (!c->rb_pinned && hipHostMalloc((void**)&c->rb_pinned, 64 * sizeof(int), hipHostMallocDefault) != hipSuccess) c->rb_pinned = nullptr;
    return c->rb_pinned;
}

static void free_buf(DevBuf& b) { if (b.p) hipFree(b.p); b.p = nullptr; b.cap = 0; }

// Host-buffer entry points are called once per frame (src/Frame.cc:467-482, src/Tracking.cc:1420, EvImBuilder.cpp:1345): their
// latency is launches and copies, not kernels.  A call lays ALL its inputs and outputs out in one device arena: the inputs are
// packed into a pinned slot and cross PCIe in ONE copy (a pageable hipMemcpy2DAsync of a 346x260 image alone cost > 1 ms), the
// outputs come back in ONE copy into a pinned landing buffer.
// upload of a call's inputs by a kernel that reads the pinned staging buffer over the link (16 bytes per thread, coalesced): for the few
// hundred KB of a one-frame call the copy engine's turn plus its hand-over to the first kernel cost more (SearchByProjection: copy 11 us +
// 12-15 us idle before the first kernel) than these reads
__global__ void arena_upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

struct Arena {
    eorb_ctx* c;
    size_t total = 0, in_end = 0;
    struct Part { const void* src; size_t off, bytes; int rows; size_t row_bytes, stride; };
    std::vector<Part> parts;
    explicit Arena(eorb_ctx* cc) : c(cc) {}
    size_t take(size_t bytes) { const size_t o = total; total = (total + std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return o; }
    // inputs first (they form the prefix that is uploaded), then reserve() for device-only / output regions
    size_t in(const void* h, size_t bytes) { const size_t o = take(bytes); if (h && bytes) parts.push_back({h, o, bytes, 0, 0, 0}); in_end = total; return o; }
    size_t in2d(const void* h, int rows, size_t row_bytes, size_t stride)
    {
        const size_t o = take((size_t)rows * row_bytes);
        if (stride == row_bytes) parts.push_back({h, o, (size_t)rows * row_bytes, 0, 0, 0});
        else parts.push_back({h, o, 0, rows, row_bytes, stride});
        in_end = total; return o;
    }
    size_t reserve(size_t bytes) { return take(bytes); }
    template <typename T> T* dev(size_t off) const { return (T*)((char*)c->arena.p + off); }
    // host_inputs: the inputs are NOT copied to the arena; the kernels that read them (once) get the pinned staging buffer itself
    // (in_ptr), which the device reads over the link.  For a few KB this saves the copy engine's turn and its hand-over to the first
    // kernel (tools/mb/call_latency.hip: 25 -> 21 us per call).  inputs_done() after the last kernel that reads them is enqueued.
    bool host_inputs = false; char* host_base = nullptr;
    template <typename T> T* in_ptr(size_t off) const { return host_inputs ? (T*)(host_base + off) : dev<T>(off); }
    void inputs_done() {}                            // (the slot stays taken until the call's wait: pinned_commit(lazy) in upload())
    int upload()
    {
        int rc = ensure(c, c->arena, total);
        if (rc) return rc;
        c->arena_gen++;                              // (whatever an earlier call left in the arena is gone)
        if (!in_end) return EORB_OK;
        char* hp = (char*)pinned(c, in_end);
        if (!hp) return set_err(c, EORB_E_HIP, "pinned alloc failed");
        for (const Part& p : parts) {
            if (p.rows) for (int r = 0; r < p.rows; r++) memcpy(hp + p.off + (size_t)r * p.row_bytes, (const char*)p.src + (size_t)r * p.stride, p.row_bytes);
            else memcpy(hp + p.off, p.src, p.bytes);
        }
        if (host_inputs) { host_base = hp; pinned_commit(c, true); return EORB_OK; }
        static const long kmax = [] { const char* e = getenv("EORB_UPLOAD_KERNEL_MAX"); return e ? atol(e) : (1L << 20); }();      // (bytes; 0: always the copy engine)
        if ((long)in_end <= kmax) {
            const size_t n16 = (in_end + 15) / 16;          // (offsets and sizes of the arena are multiples of 256; the staging buffer is at least as long)
            arena_upload_kernel<<<(unsigned)std::min<size_t>((n16 + 255) / 256, 512), 256, 0, c->stream>>>((const uint4*)hp, (uint4*)c->arena.p, n16);
            EORB_LAUNCH_CHECK(c, "arena_upload_kernel");
        } else
            EORB_HIP(c, hipMemcpyAsync(c->arena.p, hp, in_end, hipMemcpyHostToDevice, c->stream));
        pinned_commit(c, true);
        return EORB_OK;
    }
    // one D2H copy of arena[off, off + bytes) + a wait for it; returns the host view of arena offset `off` (valid until the next call).
    // download_begin / download_wait: the same in two halves -- what the caller launches in between (work the results do not depend on:
    // the LK reference kept for the next call) runs after the copy and is not waited for.
    size_t dl_off = 0;
    int download_begin(size_t off, size_t bytes)
    {
        if (c->dl_cap < bytes) {
            if (c->dl_pinned) { hipHostFree(c->dl_pinned); c->dl_pinned = nullptr; c->dl_cap = 0; }
            const size_t want = bytes + bytes / 2 + 4096;
            if (hipHostMalloc(&c->dl_pinned, want, hipHostMallocDefault) != hipSuccess) { c->dl_pinned = nullptr; return set_err(c, EORB_E_HIP, "pinned alloc failed"); }
            c->dl_cap = want;
        }
        // (the way back like the way in: up to a few hundred KB a kernel writes the pinned buffer; offsets of the arena are multiples of
        // 256 and both buffers longer than the rounded size)
        static const long kmax = [] { const char* e = getenv("EORB_DOWNLOAD_KERNEL_MAX"); return e ? atol(e) : (1L << 20); }();      // (bytes; 0: always the copy engine)
        if ((long)bytes <= kmax && !(off & 15)) {
            const size_t n16 = (bytes + 15) / 16;
            arena_upload_kernel<<<(unsigned)std::min<size_t>((n16 + 255) / 256, 512), 256, 0, c->stream>>>((const uint4*)((char*)c->arena.p + off), (uint4*)c->dl_pinned, n16);
            EORB_LAUNCH_CHECK(c, "arena download kernel");
        } else
        EORB_HIP(c, hipMemcpyAsync(c->dl_pinned, (char*)c->arena.p + off, bytes, hipMemcpyDeviceToHost, c->stream));
        if (!c->dl_event && hipEventCreateWithFlags(&c->dl_event, hipEventDisableTiming) != hipSuccess) { c->dl_event = nullptr; return set_err(c, EORB_E_HIP, "event"); }
        EORB_HIP(c, hipEventRecord(c->dl_event, c->stream));
        dl_off = off;
        return EORB_OK;
    }
    int download_wait(const char** host)
    {
        // EORB_SYNC_SPIN=1: poll instead of blocking (one 2 000-event slice through ev2im_gauss + detect: p50 0.260 ->
        // 0.237 ms, p95 0.277 -> 0.326 ms, and a CPU core kept busy: off by default)
        static const int spin = [] { const char* e = getenv("EORB_SYNC_SPIN"); return e ? atoi(e) : 0; }();
        if (spin) {
            hipError_t q;
            while ((q = hipEventQuery(c->dl_event)) == hipErrorNotReady) {}
            if (q != hipSuccess) return hip_check(c, q, "hipEventQuery");
        } else
            EORB_HIP(c, hipEventSynchronize(c->dl_event));
        pinned_release_lazy(c);
        *host = (const char*)c->dl_pinned - dl_off;
        return EORB_OK;
    }
    int download(size_t off, size_t bytes, const char** host)
    {
        int rc = download_begin(off, bytes);
        if (rc) return rc;
        if ((rc = download_wait(host))) return rc;
        // (profiling: the scopes' events are collected when the stream is idle -- eorb_prof_* synchronise before they read)
        return EORB_OK;
    }
};

}  // namespace eorb

using namespace eorb;

extern "C" {

const char* eorb_version(void) { return "eorb_fe 0.1.0 (gfx950)"; }

int eorb_create(int device, void* hip_stream, eorb_ctx** out)
{
    if (!out) return EORB_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return EORB_E_HIP;
    if (device < 0 || device >= ndev) return EORB_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return EORB_E_HIP;
    eorb_ctx* c = new eorb_ctx();
    c->device = device;
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return EORB_E_HIP; }
        c->own_stream = true;
    }
    if (ensure(c, c->status, 64) != EORB_OK || hipMemsetAsync(c->status.p, 0, 64, c->stream) != hipSuccess) { eorb_destroy(c); return EORB_E_HIP; }
    *out = c;
    return EORB_OK;
}

void eorb_destroy(eorb_ctx* c)
{
    if (!c) return;
    fe_enter(c);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    DevBuf* bufs[] = {&c->ev16, &c->chunks, &c->segoff, &c->entries, &c->img_f32, &c->img_u8, &c->minmax, &c->tile_order, &c->order_hist, &c->lut, &c->src_info, &c->stamps, &c->sl_tab, &c->sl_tile, &c->sl_rows, &c->sl_trace, &c->dd_tab, &c->dd_src_info, &c->dd_stamps, &c->dd_sl_tab, &c->dd_sl_tile, &c->dd_sl_rows, &c->dd_ev, &c->dd_cnt, &c->focus_sd, &c->voc, &c->klt_pyr, &c->klt_der, &c->klt_scratch, &c->pyr, &c->score,
                      &c->blur, &c->cell_cnt, &c->cell_cand, &c->lvl_cnt, &c->lvl_kp, &c->kp_angle, &c->out_kp, &c->out_desc,
                      &c->out_oob, &c->out_n, &c->oct_scratch, &c->in_img, &c->m_a, &c->m_b, &c->m_c, &c->m_d, &c->m_e, &c->m_f,
                      &c->m_g, &c->m_h, &c->m_i, &c->m_j, &c->fe_prev_kp, &c->fe_prev_desc, &c->fe_prev_n, &c->fe_pm,
                      &c->orb.tabs, &c->orb.geom, &c->status, &c->win_ws, &c->win_total, &c->arena, &c->l1_ref_img, &c->l1_ref_pts, &c->ev_info, &c->ev_stamps, &c->pd_hash, &c->pd_lut, &c->pd_src_info, &c->pd_sl_tab, &c->pd_sl_tile, &c->pd_sl_rows, &c->pd_cnt};
    for (DevBuf* b : bufs) free_buf(*b);
    for (auto& s : c->pinned) { if (s.ev) hipEventDestroy(s.ev); if (s.p) hipHostFree(s.p); }
    for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
    if (c->dl_pinned) hipHostFree(c->dl_pinned);
    if (c->dl_event) hipEventDestroy(c->dl_event);
    if (c->rb_pinned) hipHostFree(c->rb_pinned);
    if (c->up_pin.p) hipHostFree(c->up_pin.p);
    if (c->dn_pin.p) hipHostFree(c->dn_pin.p);
    for (hipEvent_t e : c->sl_ev) if (e) hipEventDestroy(e);
    if (c->sl_side) hipStreamDestroy(c->sl_side);
    if (c->sl_pstream) hipStreamDestroy(c->sl_pstream);
    if (c->sl_gstream) hipStreamDestroy(c->sl_gstream);
    for (auto& w : c->sl_ws) { free_buf(w.chunks); free_buf(w.segoff); free_buf(w.entries); free_buf(w.tile_order); free_buf(w.plan); free_buf(w.hot); free_buf(w.rec16); }
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

int eorb_sync(eorb_ctx* c)
{
    if (!c) return EORB_E_ARG;
    fe_enter(c);
    EORB_HIP(c, fe_stream_sync(c));
    // sticky status of the asynchronous (*_dev) paths: kernels OR their overflow bits into a device word; report it once
    int32_t bits = 0;
    EORB_HIP(c, hipMemcpyAsync(&bits, c->status.p, sizeof(bits), hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    if (bits) {
        EORB_HIP(c, hipMemsetAsync(c->status.p, 0, sizeof(bits), c->stream));
        if (bits & 256)
            return set_err(c, EORB_E_HIP, "internal error: the slot gather found its LDS rows at a non-zero base (status %d); the images of that call were not written", bits);
        return set_err(c, EORB_E_CAPACITY, "a batched call exceeded an internal capacity (octree flags %d: 1 = candidates, 2 = node pool / "
                       "size list, 4 = keypoints per level); its keypoints are truncated", bits);
    }
    return EORB_OK;
}

int eorb_debug_option(eorb_ctx* c, const char* name, int value)
{
    if (!c || !name) return EORB_E_ARG;
    if (!strcmp(name, "octree_pool_shrink")) { c->dbg_pool_shrink = value; return EORB_OK; }
    if (!strcmp(name, "octree_force_global")) { c->dbg_force_global = value; return EORB_OK; }
    if (!strcmp(name, "octree_list_algorithm")) { c->dbg_oct_list = value; return EORB_OK; }
    if (!strcmp(name, "win_list_cap")) { c->dbg_win_wcap = value; return EORB_OK; }
    if (!strcmp(name, "win_pool_cap")) { c->dbg_win_ecap = value; return EORB_OK; }
    if (!strcmp(name, "orb_three_launches")) { c->dbg_orb_three_launches = value; return EORB_OK; }
    if (!strcmp(name, "win_lds_entries")) { c->dbg_win_lds_ents = value; return EORB_OK; }
    if (!strcmp(name, "gather_form")) { c->dbg_gather_form = value; return EORB_OK; }
    if (!strcmp(name, "dedupe_min_events")) { c->dbg_dd_min = value; return EORB_OK; }
    if (!strcmp(name, "slot_rank")) { c->dbg_slot_rank = value; return EORB_OK; }
    if (!strcmp(name, "slot_hot_min")) { c->dbg_slot_hot_min = value; return EORB_OK; }
    if (!strcmp(name, "slot_hot_cap")) { c->dbg_slot_hot_cap = value; return EORB_OK; }
    if (!strcmp(name, "slot_halves")) { c->dbg_slot_halves = value; return EORB_OK; }
    if (!strcmp(name, "position_dict")) { c->dbg_pd = value; if (!value) { c->pd_valid = 0; c->dd_keep = 0; } return EORB_OK; }
    if (!strcmp(name, "slot_hot_waves")) { c->dbg_slot_hot_waves = value; return EORB_OK; }
    if (!strcmp(name, "slot_prerank")) { c->dbg_slot_prerank = value; return EORB_OK; }
    return set_err(c, EORB_E_ARG, "debug option '%s' unknown", name);
}

long long eorb_debug_counter(eorb_ctx* c, const char* name)
{
    if (!c || !name) return -1;
    if (!strcmp(name, "slot_calls")) return c->sl_calls;
    if (!strcmp(name, "slot_rank_ok")) return c->sl_rank_ok;
    if (!strcmp(name, "dict_hits")) return c->pd_hits;                    // bulk float calls served by the frozen position dictionary
    if (!strcmp(name, "dict_misses")) return c->pd_misses;                // ... that found a new position and tabulated afresh
    if (!strcmp(name, "dict_positions")) return c->pd_valid ? c->pd_K : 0;
    if (!strcmp(name, "slot_scatter_form")) return c->sl_last_rank;       // the scatter of the last slot-form call: 1 rank form, 0 ballot form
    if (!strcmp(name, "slot_chunk")) return c->sl_last_chunk;
    if (!strcmp(name, "slot_parts")) return c->sl_last_parts;              // 2: the last slot-form call ran its batch as two halves
    if (!strcmp(name, "slot_hot_overflow")) {           // lists the last slot-form call handed back to the LDS gather because their length bucket was full (synchronises)
        long long n = 0;
        for (int part = 0; part < std::max(c->sl_last_parts, 1); part++) {
            if (!c->sl_ws[part].hot.p) continue;
            uint32_t h = 0;
            if (hipMemcpyAsync(&h, (uint32_t*)c->sl_ws[part].hot.p + 33, sizeof(h), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
            n += h;
        }
        return n;
    }
    if (!strcmp(name, "oct_lds_only")) return c->orb.oct_all_lds[0] | (c->orb.oct_all_lds[1] << 1);      // octree working set entirely in LDS: single frames | batches
    if (!strcmp(name, "oct_lds_bytes")) return c->orb.oct_lds[0];
    if (!strcmp(name, "oct_dynamic")) return c->orb.oct_dyn[0] != 0 ? 1 : 0;          // single frames use the dynamic LDS placement
    if (!strcmp(name, "oct_redo_levels")) {                                            // levels of the last single-frame extraction that did not fit it (synchronises)
        const int nl = c->orb.nlevels;
        if (!c->orb.oct_dyn[0] || !c->lvl_cnt.p || nl <= 0) return 0;
        std::vector<int32_t> f((size_t)nl);
        if (hipMemcpyAsync(f.data(), (const char*)c->lvl_cnt.p + (size_t)nl * sizeof(int32_t) + 64, sizeof(int32_t) * (size_t)nl, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) return -1;
        long long r = 0; for (int v : f) r += v ? 1 : 0;
        return r;
    }
    if (!strcmp(name, "oct_direct_cap")) return c->orb.oct_direct_cap[0] | ((long long)c->orb.oct_direct_cap[1] << 16);
    if (!strcmp(name, "oct_scratch_bytes")) return c->orb.oct_scratch[0];
    if (!strcmp(name, "slot_hot_items")) {              // lists the last slot-form call handed to the register-row kernel (synchronises)
        long long n = 0;
        for (int part = 0; part < std::max(c->sl_last_parts, 1); part++) {
            if (!c->sl_ws[part].hot.p) continue;
            uint32_t h[16];
            if (hipMemcpyAsync(h, c->sl_ws[part].hot.p, sizeof(h), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
            for (int i = 0; i < 16; i++) n += h[i];      // (sl_tasks_kernel capped the counts at the buckets' capacity)
        }
        return n;
    }
    if (!strcmp(name, "slot_entries") || !strcmp(name, "slot_hot_entries")) {
        // list entries the last slot-form call's gather kernels walked: all of them / those of the lists handed to the register-row kernel
        // (synchronises; reads the scan's per-(slice, tile) counts and the hot descriptors back)
        const bool hot = name[5] == 'h';
        if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
        long long n = 0;
        for (int part = 0; part < std::max(c->sl_last_parts, 1); part++) {
            const eorb_ctx::SlotWS& w = c->sl_ws[part];
            if (hot) {
                if (!w.hot.p) continue;
                uint32_t cnt[16];
                if (hipMemcpy(cnt, w.hot.p, sizeof(cnt), hipMemcpyDeviceToHost) != hipSuccess) return -1;
                for (int b = 0; b < 16; b++) {
                    std::vector<uint32_t> d(8 * (size_t)cnt[b]);
                    if (cnt[b] && hipMemcpy(d.data(), (const char*)w.hot.p + 256 + (size_t)b * 8192 * 32, 32 * (size_t)cnt[b], hipMemcpyDeviceToHost) != hipSuccess) return -1;
                    for (uint32_t k = 0; k < cnt[b]; k++) n += d[8 * (size_t)k + 2];
                }
            } else {
                if (!w.tile_order.p || !c->sl_last_nb[part]) continue;
                std::vector<uint32_t> t((size_t)c->sl_last_nb[part]);
                if (hipMemcpy(t.data(), w.tile_order.p, sizeof(uint32_t) * t.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
                for (uint32_t v : t) n += v;
            }
        }
        return n;
    }
    if (!strcmp(name, "slot_flags")) {
        if (!c->sl_tile.p || !c->sl_info_off) return 0;
        int flags = 0;
        if (hipMemcpyAsync(&flags, (char*)c->sl_tile.p + c->sl_info_off + 3 * sizeof(int), sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
        return flags;
    }
    return -1;
}

const char* eorb_last_error(eorb_ctx* c) { return c ? c->err.c_str() : "null context"; }

int eorb_prof_enable(eorb_ctx* c, int on) { if (!c) return EORB_E_ARG; c->prof = on != 0; return EORB_OK; }
int eorb_prof_only(eorb_ctx* c, const char* names)
{
    if (!c) return EORB_E_ARG;
    c->prof_only = (names && *names) ? std::string(",") + names + "," : std::string();
    return EORB_OK;
}
int eorb_prof_reset(eorb_ctx* c)
{
    if (!c) return EORB_E_ARG;
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    c->profs.clear();
    return EORB_OK;
}
int eorb_prof_count(eorb_ctx* c) { if (!c) return EORB_E_ARG; hipStreamSynchronize(c->stream); prof_collect(c); return (int)c->profs.size(); }
int eorb_prof_get(eorb_ctx* c, int i, const char** name, double* total_ms, int64_t* launches)
{
    if (!c || i < 0 || i >= (int)c->profs.size()) return EORB_E_ARG;
    if (name) *name = c->profs[i].name.c_str();
    if (total_ms) *total_ms = c->profs[i].total_ms;
    if (launches) *launches = c->profs[i].launches;
    return EORB_OK;
}

void eorb_pack_events(const eorb_event* ev, size_t n, eorb_event16* out)
{
    for (size_t i = 0; i < n; i++) {
        out[i].x = ev[i].x; out[i].y = ev[i].y;
        double t = ev[i].ts < 0 ? 0.0 : ev[i].ts;
        uint64_t u; memcpy(&u, &t, 8);
        if (!ev[i].p) u |= 0x8000000000000000ull;
        memcpy(&out[i].t, &u, 8);
    }
}

void* eorb_dev_alloc(eorb_ctx* c, size_t bytes)
{
    if (!c) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { set_err(c, EORB_E_HIP, "hipMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
int eorb_dev_free(eorb_ctx* c, void* p) { if (!c) return EORB_E_ARG; hipStreamSynchronize(c->stream); EORB_HIP(c, hipFree(p)); return EORB_OK; }
int eorb_dev_upload(eorb_ctx* c, void* d, const void* h, size_t bytes)
{
    if (!c) return EORB_E_ARG;
    EORB_HIP(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}
int eorb_dev_download(eorb_ctx* c, void* h, const void* d, size_t bytes)
{
    if (!c) return EORB_E_ARG;
    EORB_HIP(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

// ---- event accumulation, host buffers -------------------------------------------------------------
static int ev_host_common(eorb_ctx* c, const eorb_event* ev, size_t n, int W, int H, float sigma, int pol, int normalized,
                          int mode_count, float* out_f32, uint8_t* out_u8, float* minmax, int* is_u8,
                          const eorb_raw_event* rawev = nullptr)
{
    if (!c) return EORB_E_ARG;
    const int raw = rawev != nullptr;
    if (raw && !c->lut_w) return set_err(c, EORB_E_NOTCONF, "ev2im_raw: eorb_set_undistort_maps not called");
    if (raw) ev = nullptr;
    if (W <= 0 || H <= 0 || (n && !ev && !raw)) return set_err(c, EORB_E_ARG, "ev2im: bad arguments");
    fe_enter(c);
    int rc;
    const size_t npix = (size_t)W * H;
    static_assert(sizeof(eorb_raw_event) == sizeof(eorb_event16), "raw and packed events share the 16-byte slot");
    std::vector<eorb_event16> packed;
    if (n && raw) {
        for (size_t i = 0; i < n; i++)
            if ((int)rawev[i].x >= c->lut_w || (int)rawev[i].y >= c->lut_h)       // the reference asserts (MyCalibrator.cpp:176)
                return set_err(c, EORB_E_ARG, "ev2im_raw: event %zu at (%u,%u) lies outside the %dx%d maps", i, rawev[i].x, rawev[i].y,
                               c->lut_w, c->lut_h);
    } else if (n) {
        packed.resize(n);
        eorb_pack_events(ev, n, packed.data());
    }
    Arena A(c);
    // (a live slice goes to the binning-free form, which reads every event once: in place, from the pinned staging buffer)
    static const int zc_env = [] { const char* e = getenv("EORB_SLICE_ZERO_COPY"); return e ? atoi(e) : 1; }();      // (A/B runs)
    A.host_inputs = zc_env != 0 && !mode_count && n > 0 && n <= 16384 && c->dbg_gather_form == 0;
    const size_t o_ev = A.in(raw ? (const void*)rawev : (const void*)packed.data(), sizeof(eorb_event16) * n);
    // outputs, contiguous: min/max (encoded | decoded) | u8 image | f32 image
    const size_t o_mm = A.reserve(64), o_u8 = A.reserve(npix), o_f32 = A.reserve(sizeof(float) * npix);
    if ((rc = A.upload())) return rc;
    int64_t offs[2] = {0, (int64_t)n};
    uint32_t* mm = A.dev<uint32_t>(o_mm);
    uint8_t* d_u8 = A.dev<uint8_t>(o_u8);
    float* d_f32 = A.dev<float>(o_f32);
    if (out_u8 && mode_count) EORB_HIP(c, hipMemsetAsync(d_u8, 0, npix, c->stream));     // count images stay empty when max == min
    const long long slot_calls0 = c->sl_calls;
    rc = ev_accumulate_dev(c, A.in_ptr<void>(o_ev), raw, offs, 1, W, H, sigma, pol, mode_count, d_f32, d_u8, normalized, mm);
    A.inputs_done();
    if (rc) return rc;
    // the slot form reports an internal fault (its gather found no rows at LDS offset 0 and wrote no image) through the sticky status
    // word: this call's copy of it travels with the outputs (word 2 of the min/max block)
    const bool slot_ran = c->sl_calls != slot_calls0;
    if (slot_ran) EORB_HIP(c, hipMemcpyAsync(mm + 2, c->status.p, 4, hipMemcpyDeviceToDevice, c->stream));
    const size_t end = out_f32 ? o_f32 + sizeof(float) * npix : (out_u8 ? o_u8 + npix : o_mm + 64);
    const char* h;
    if ((rc = A.download(o_mm, end - o_mm, &h))) return rc;
    if (slot_ran) {
        int32_t st; memcpy(&st, h + o_mm + 8, 4);
        if (st & 256) return set_err(c, EORB_E_HIP, "internal error: the slot gather found its LDS rows at a non-zero base; no image was written");
    }
    // the running extremes come back in their order-preserving integer encoding (enc_f32 of the gather kernels): decoded here
    float hmm[2];
    for (int k = 0; k < 2; k++) {
        uint32_t e; memcpy(&e, h + o_mm + 4 * k, 4);
        const uint32_t u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
        memcpy(&hmm[k], &u, 4);
    }
    if (out_f32) memcpy(out_f32, h + o_f32, sizeof(float) * npix);
    if (out_u8) memcpy(out_u8, h + o_u8, npix);
    if (minmax) { minmax[0] = hmm[0]; minmax[1] = hmm[1]; }
    if (is_u8) *is_u8 = mode_count ? (normalized && hmm[1] > hmm[0]) : (normalized != 0);
    return EORB_OK;
}

int eorb_ev2im(eorb_ctx* c, const eorb_event* ev, size_t n, int W, int H, int pol, int normalized,
               float* out_f32, uint8_t* out_u8, float* minmax, int* is_u8)
{
    return ev_host_common(c, ev, n, W, H, 0.f, pol, normalized, 1, out_f32, out_u8, minmax, is_u8);
}

int eorb_ev2im_gauss(eorb_ctx* c, const eorb_event* ev, size_t n, int W, int H, float sigma, int pol,
                     int normalized, float* out_f32, uint8_t* out_u8, float* minmax)
{
    if (c && !(sigma > 0.f)) return set_err(c, EORB_E_ARG, "ev2im_gauss: sigma must be > 0");
    return ev_host_common(c, ev, n, W, H, sigma, pol, normalized, 0, out_f32, out_u8, minmax, nullptr);
}

static int up(eorb_ctx* c, DevBuf& b, const void* h, size_t bytes);

// ---- raw sensor events + undistortion maps ----------------------------------------------------------
int eorb_set_undistort_maps(eorb_ctx* c, const float* mapX, const float* mapY, int LW, int LH, int checkInImage)
{
    if (!c) return EORB_E_ARG;
    if (!mapX || !mapY || LW <= 0 || LH <= 0 || LW > 65535 || LH > 65535 || (int64_t)LW * LH >= (1ll << 31))
        return set_err(c, EORB_E_ARG, "set_undistort_maps: bad arguments");
    fe_enter(c);
    const size_t n = (size_t)LW * LH;
    std::vector<float> xy(2 * n);
    for (size_t i = 0; i < n; i++) { xy[2 * i] = mapX[i]; xy[2 * i + 1] = mapY[i]; }
    int rc;
    if ((rc = up(c, c->lut, xy.data(), sizeof(float) * 2 * n))) return rc;
    if ((rc = up_flush(c))) return rc;
    EORB_HIP(c, fe_stream_sync(c));
    c->lut_w = LW; c->lut_h = LH; c->lut_check = checkInImage != 0;
    c->lut_key_W = c->lut_key_H = c->lut_key_mode = -1; c->lut_key_sigma = -1.f;      // derived tables are stale
    return EORB_OK;
}

int eorb_undistort_events(eorb_ctx* c, const eorb_raw_event* raw, size_t n, int W, int H, double tsFactor, eorb_event* out, size_t* n_out)
{
    if (!c) return EORB_E_ARG;
    if (!c->lut_w) return set_err(c, EORB_E_NOTCONF, "undistort_events: eorb_set_undistort_maps not called");
    if ((n && (!raw || !out)) || !n_out || W <= 0 || H <= 0) return set_err(c, EORB_E_ARG, "undistort_events: bad arguments");
    fe_enter(c);
    *n_out = 0;
    if (!n) return EORB_OK;
    for (size_t i = 0; i < n; i++)
        if ((int)raw[i].x >= c->lut_w || (int)raw[i].y >= c->lut_h)
            return set_err(c, EORB_E_ARG, "undistort_events: event %zu at (%u,%u) lies outside the %dx%d maps", i, raw[i].x, raw[i].y,
                           c->lut_w, c->lut_h);
    int rc;
    if ((rc = up(c, c->ev16, raw, sizeof(eorb_raw_event) * n))) return rc;
    if ((rc = up_flush(c))) return rc;
    if ((rc = ensure(c, c->entries, sizeof(eorb_event) * n))) return rc;
    const int nblk = (int)((n + 1023) / 1024);
    if ((rc = ensure(c, c->segoff, sizeof(uint32_t) * ((size_t)nblk + 2)))) return rc;
    if ((rc = ev_undistort_dev(c, (const eorb_raw_event*)c->ev16.p, n, W, H, tsFactor, (eorb_event*)c->entries.p, (uint32_t*)c->segoff.p))) return rc;
    uint32_t kept = 0;
    EORB_HIP(c, hipMemcpyAsync(&kept, (uint32_t*)c->segoff.p + nblk, 4, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    if (kept) EORB_HIP(c, hipMemcpyAsync(out, c->entries.p, sizeof(eorb_event) * kept, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    *n_out = kept;
    return EORB_OK;
}

int eorb_parse_events_text(eorb_ctx* c, const char* text, size_t nbytes, eorb_raw_event* out, size_t cap, size_t* n_out,
                           int64_t* bad_line)
{
    if (!c) return EORB_E_ARG;
    if ((nbytes && !text) || !n_out || (cap && !out)) return set_err(c, EORB_E_ARG, "parse_events_text: bad arguments");
    fe_enter(c);
    *n_out = 0; if (bad_line) *bad_line = -1;
    if (!nbytes) return EORB_OK;
    // every line of the accepted grammar is at least 8 bytes ("0 0 0 0\n"); shorter ones are comments / blanks or errors, so
    // the line capacity is bounded by the caller's event capacity plus what the text could hold otherwise
    const size_t max_lines = nbytes / 2 + 2;
    int rc;
    if ((rc = up(c, c->in_img, text, nbytes))) return rc;
    if ((rc = up_flush(c))) return rc;
    const size_t nblk = (nbytes + 1023) / 1024;
    // workspaces: lineend u64 | parsed events | status | block sums
    if ((rc = ensure(c, c->entries, sizeof(uint64_t) * max_lines))) return rc;
    if ((rc = ensure(c, c->ev16, sizeof(eorb_raw_event) * max_lines))) return rc;
    if ((rc = ensure(c, c->segoff, max_lines + 64))) return rc;
    if ((rc = ensure(c, c->tile_order, sizeof(uint32_t) * (std::max(nblk, (max_lines + 1023) / 1024) + 8)))) return rc;
    if ((rc = ensure(c, c->chunks, sizeof(eorb_raw_event) * max_lines))) return rc;
    uint32_t res[3];
    if ((rc = ev_parse_text_dev(c, (const char*)c->in_img.p, nbytes, (uint64_t*)c->entries.p, (eorb_raw_event*)c->ev16.p,
                                (uint8_t*)c->segoff.p, (eorb_raw_event*)c->chunks.p, (uint32_t*)c->tile_order.p, max_lines, res))) return rc;
    if (res[2] != 0xffffffffu) {
        if (bad_line) *bad_line = (int64_t)res[2];
        return set_err(c, EORB_E_ARG, "parse_events_text: line %u is outside the accepted \"ts x y p\" grammar", res[2]);
    }
    if (res[1] > cap) return set_err(c, EORB_E_CAPACITY, "parse_events_text: %u events, room for %zu", res[1], cap);
    if (res[1]) EORB_HIP(c, hipMemcpyAsync(out, c->chunks.p, sizeof(eorb_raw_event) * (size_t)res[1], hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    *n_out = res[1];
    return EORB_OK;
}

int eorb_ev2im_gauss_raw(eorb_ctx* c, const eorb_raw_event* raw, size_t n, int W, int H, float sigma, int pol, int normalized,
                         float* out_f32, uint8_t* out_u8, float* minmax)
{
    if (c && !(sigma > 0.f)) return set_err(c, EORB_E_ARG, "ev2im_gauss_raw: sigma must be > 0");
    if (c && n && !raw) return set_err(c, EORB_E_ARG, "ev2im_gauss_raw: bad arguments");
    static const eorb_raw_event none{};
    return ev_host_common(c, nullptr, n, W, H, sigma, pol, normalized, 0, out_f32, out_u8, minmax, nullptr, raw ? raw : &none);
}

int eorb_ev2im_raw(eorb_ctx* c, const eorb_raw_event* raw, size_t n, int W, int H, int pol, int normalized,
                   float* out_f32, uint8_t* out_u8, float* minmax, int* is_u8)
{
    if (c && n && !raw) return set_err(c, EORB_E_ARG, "ev2im_raw: bad arguments");
    static const eorb_raw_event none{};
    return ev_host_common(c, nullptr, n, W, H, 0.f, pol, normalized, 1, out_f32, out_u8, minmax, is_u8, raw ? raw : &none);
}

// ---- motion-compensated accumulation (f1) ----
static int mci_common(eorb_ctx* c, const eorb_event* ev, size_t n, const eorb_camera* cam, int se3, double angle, const double* axis,
                      const double* t, float medDepth, const float* depth, const float* params, int nparams, int W, int H,
                      float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax)
{
    if (!c) return EORB_E_ARG;
    if (W <= 0 || H <= 0 || (n && !ev) || !cam || !(sigma > 0.f)) return set_err(c, EORB_E_ARG, "ev2mci: bad arguments");
    fe_enter(c);
    const size_t npix = (size_t)W * H;
    if (n == 0) {                         // "no events" -> zero CV_32FC1 image (:292-295)
        if (out_f32) memset(out_f32, 0, sizeof(float) * npix);
        if (out_u8) memset(out_u8, 0, npix);
        if (minmax) { minmax[0] = 0.f; minmax[1] = -1000000.0f; }
        return EORB_OK;
    }
    if (n > 0x7fffffff) return set_err(c, EORB_E_CAPACITY, "ev2mci: too many events");
    int rc;
    if ((rc = ensure(c, c->ev16, sizeof(eorb_event16) * n))) return rc;
    if ((rc = ensure(c, c->m_a, sizeof(eorb_event16) * n))) return rc;
    if ((rc = ensure(c, c->img_f32, sizeof(float) * npix))) return rc;
    if ((rc = ensure(c, c->img_u8, npix))) return rc;
    if ((rc = ensure(c, c->minmax, 64))) return rc;
    std::vector<eorb_event16> packed(n);
    eorb_pack_events(ev, n, packed.data());
    EORB_HIP(c, hipMemcpyAsync(c->m_a.p, packed.data(), sizeof(eorb_event16) * n, hipMemcpyHostToDevice, c->stream));
    const float* d_depth = nullptr;
    if (depth) {
        if ((rc = up(c, c->m_b, depth, sizeof(float) * n))) return rc;
        if ((rc = up_flush(c))) return rc;
        d_depth = (const float*)c->m_b.p;
    }
    EORB_HIP(c, fe_stream_sync(c));
    if (cam->model != 0 && cam->model != 1) return set_err(c, EORB_E_ARG, "ev2mci: camera model %d unknown", cam->model);
    if (se3) rc = ev_warp_se3_dev(c, (const eorb_event16*)c->m_a.p, (eorb_event16*)c->ev16.p, (int)n, cam, angle, axis, t, medDepth, d_depth);
    else rc = ev_warp_se2_dev(c, (const eorb_event16*)c->m_a.p, (eorb_event16*)c->ev16.p, (int)n, cam, params, nparams);
    if (rc) return rc;
    int64_t offs[2] = {0, (int64_t)n};
    uint32_t* mm = (uint32_t*)c->minmax.p;
    float* mmf = (float*)((char*)c->minmax.p + 16);
    // (warped events: no two share a position -- the position table of the bulk float form would only be filled and thrown away)
    const int64_t dd_saved = c->dbg_dd_min; c->dbg_dd_min = 0;
    rc = ev_accumulate_dev(c, c->ev16.p, 0, offs, 1, W, H, sigma, pol, 0, (float*)c->img_f32.p, (uint8_t*)c->img_u8.p,
                           normalized, mm);
    c->dbg_dd_min = dd_saved;
    if (rc) return rc;
    if ((rc = ev_decode_minmax(c, mm, mmf, 1))) return rc;
    float hmm[2];
    EORB_HIP(c, hipMemcpyAsync(hmm, mmf, 8, hipMemcpyDeviceToHost, c->stream));
    if (out_f32) EORB_HIP(c, hipMemcpyAsync(out_f32, c->img_f32.p, sizeof(float) * npix, hipMemcpyDeviceToHost, c->stream));
    if (out_u8 && normalized) EORB_HIP(c, hipMemcpyAsync(out_u8, c->img_u8.p, npix, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    if (minmax) { minmax[0] = hmm[0]; minmax[1] = hmm[1]; }
    return EORB_OK;
}

static eorb_camera pinhole_cam(const eorb_pinhole* p) { eorb_camera c{}; if (p) { c.fx = p->fx; c.fy = p->fy; c.cx = p->cx; c.cy = p->cy; } return c; }

int eorb_ev2mci_se3_cam(eorb_ctx* c, const eorb_event* ev, size_t n, const eorb_camera* cam, double angle, const double axis[3],
                        const double t[3], float medDepth, const float* depth_per_event, int W, int H, float sigma, int pol,
                        int normalized, float* out_f32, uint8_t* out_u8, float* minmax)
{
    if (c && (!axis || !t)) return set_err(c, EORB_E_ARG, "ev2mci_se3: null pose");
    return mci_common(c, ev, n, cam, 1, angle, axis, t, medDepth, depth_per_event, nullptr, 0, W, H, sigma, pol, normalized, out_f32, out_u8, minmax);
}

int eorb_ev2mci_se2_cam(eorb_ctx* c, const eorb_event* ev, size_t n, const eorb_camera* cam, const float* params2D, int nparams,
                        int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax)
{
    if (c && (!params2D || nparams < 3)) return set_err(c, EORB_E_ARG, "ev2mci_se2: need at least 3 parameters");
    return mci_common(c, ev, n, cam, 0, 0.0, nullptr, nullptr, 0.f, nullptr, params2D, nparams, W, H, sigma, pol, normalized, out_f32, out_u8, minmax);
}

int eorb_ev2mci_se3(eorb_ctx* c, const eorb_event* ev, size_t n, const eorb_pinhole* cam, double angle, const double axis[3],
                    const double t[3], float medDepth, const float* depth_per_event, int W, int H, float sigma, int pol,
                    int normalized, float* out_f32, uint8_t* out_u8, float* minmax)
{
    const eorb_camera cc = pinhole_cam(cam);
    return eorb_ev2mci_se3_cam(c, ev, n, cam ? &cc : nullptr, angle, axis, t, medDepth, depth_per_event, W, H, sigma, pol, normalized, out_f32, out_u8, minmax);
}

int eorb_ev2mci_se2(eorb_ctx* c, const eorb_event* ev, size_t n, const eorb_pinhole* cam, const float* params2D, int nparams,
                    int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax)
{
    const eorb_camera cc = pinhole_cam(cam);
    return eorb_ev2mci_se2_cam(c, ev, n, cam ? &cc : nullptr, params2D, nparams, W, H, sigma, pol, normalized, out_f32, out_u8, minmax);
}

int eorb_measure_image_focus_n(eorb_ctx* c, const float* imgs, int n, int W, int H, float* focus)
{
    if (!c) return EORB_E_ARG;
    if (!imgs || !focus || W <= 0 || H <= 0 || n < 1 || n > 64) return set_err(c, EORB_E_ARG, "measure_image_focus: bad arguments");
    fe_enter(c);
    int rc;
    if ((rc = up(c, c->img_f32, imgs, sizeof(float) * (size_t)W * H * n))) return rc;
    if ((rc = up_flush(c))) return rc;
    if ((rc = ensure(c, c->minmax, 256))) return rc;
    if ((rc = ev_focus_dev(c, (const float*)c->img_f32.p, n, W, H, (float*)c->minmax.p))) return rc;
    EORB_HIP(c, hipMemcpyAsync(focus, c->minmax.p, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}
int eorb_measure_image_focus(eorb_ctx* c, const float* img, int W, int H, float* focus) { return eorb_measure_image_focus_n(c, img, 1, W, H, focus); }

int eorb_normalize_minmax_u8(eorb_ctx* c, const float* img, int W, int H, uint8_t* out)
{
    if (!c) return EORB_E_ARG;
    if (!img || !out || W <= 0 || H <= 0) return set_err(c, EORB_E_ARG, "normalize_minmax_u8: bad arguments");
    fe_enter(c);
    const size_t npix = (size_t)W * H;
    int rc;
    if ((rc = up(c, c->img_f32, img, sizeof(float) * npix))) return rc;
    if ((rc = up_flush(c))) return rc;
    if ((rc = ensure(c, c->img_u8, npix))) return rc;
    if ((rc = ensure(c, c->minmax, 64))) return rc;
    if ((rc = ev_cvnormalize_dev(c, (const float*)c->img_f32.p, (int)npix, (uint32_t*)c->minmax.p, (uint8_t*)c->img_u8.p))) return rc;
    EORB_HIP(c, hipMemcpyAsync(out, c->img_u8.p, npix, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

// ---- the L1 image builder's per-chunk path, one call per chunk (src/Event/EvImBuilder.cpp:1300-1515) ----------------------------
// resolveMinMaxVals' start values (min 0, max -1e6: src/Event/EventConversion.cc:224-225) in the order-preserving encoding of the
// gather kernels' atomics (enc_f32), uploaded with a call's events instead of being written by a kernel
static const uint32_t kMinMaxPreset[16] = {0x80000000u, 0x368bdbffu, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// events of a chunk -> arena; float EventData are packed to the 16-byte record on the host, raw sensor events go as they are
static int slice_events_in(eorb_ctx* c, Arena& A, const eorb_event* ev, const eorb_raw_event* raw, size_t n, std::vector<eorb_event16>& packed,
                           size_t* o_ev, int* is_raw, const char* who)
{
    if (n && !ev && !raw) return set_err(c, EORB_E_ARG, "%s: no events", who);
    if (ev && raw) return set_err(c, EORB_E_ARG, "%s: float events OR raw sensor events", who);
    *is_raw = raw != nullptr;
    if (raw) {
        if (!c->lut_w) return set_err(c, EORB_E_NOTCONF, "%s: raw events need eorb_set_undistort_maps first", who);
        for (size_t i = 0; i < n; i++)
            if ((int)raw[i].x >= c->lut_w || (int)raw[i].y >= c->lut_h)       // the reference asserts (MyCalibrator.cpp:176)
                return set_err(c, EORB_E_ARG, "%s: event %zu at (%u,%u) lies outside the %dx%d maps", who, i, raw[i].x, raw[i].y, c->lut_w, c->lut_h);
        *o_ev = A.in(raw, sizeof(eorb_raw_event) * n);
    } else {
        packed.resize(n);
        if (n) eorb_pack_events(ev, n, packed.data());
        *o_ev = A.in(packed.data(), sizeof(eorb_event16) * n);
    }
    return EORB_OK;
}

int eorb_ev_slice_extract(eorb_ctx* c, const eorb_event* ev, const eorb_raw_event* raw, size_t n, float sigma, int lap0, int lap1,
                          int want_desc, eorb_keypoint* kps, uint8_t* desc, uint8_t* oob, int cap, int* n_out, int* mono_index,
                          uint8_t* out_u8)
{
    if (!c) return EORB_E_ARG;
    if (n_out) *n_out = 0;
    OrbState& o = c->orb;
    if (!o.configured) return set_err(c, EORB_E_NOTCONF, "ev_slice_extract: eorb_orb_configure not called");
    if (!(sigma > 0.f)) return set_err(c, EORB_E_ARG, "ev_slice_extract: sigma must be > 0");
    fe_enter(c);
    const int W = o.W, H = o.H;
    const size_t npix = (size_t)W * H, mo = (size_t)o.max_out;
    int rc, is_raw = 0;
    Arena A(c);
    std::vector<eorb_event16> packed;
    size_t o_ev = 0;
    if ((rc = slice_events_in(c, A, ev, raw, n, packed, &o_ev, &is_raw, "ev_slice_extract"))) return rc;
    // the running extremes travel initialised with the events (no launch for them); device-only: float image; outputs, contiguous:
    // u8 image | {n, mono, flag, pad} | keypoints | descriptors | oob
    // a live slice (the binning-free form reads every event once, in ev_pre_kernel): the events stay in pinned host memory and the
    // extremes are initialised by that kernel -- no copy in front of the first launch
    static const int zc_env = [] { const char* e = getenv("EORB_SLICE_ZERO_COPY"); return e ? atoi(e) : 1; }();      // (A/B runs)
    const bool zc = zc_env != 0 && n > 0 && n <= 16384 && c->dbg_gather_form == 0;
    A.host_inputs = zc;
    const size_t o_mm = zc ? A.reserve(sizeof(kMinMaxPreset)) : A.in(kMinMaxPreset, sizeof(kMinMaxPreset)), o_f32 = A.reserve(sizeof(float) * npix);
    const size_t o_u8 = A.reserve(npix), o_n = A.reserve(16), o_kp = A.reserve(sizeof(eorb_keypoint) * mo), o_desc = A.reserve(32 * mo), o_oob = A.reserve(mo);
    if ((rc = A.upload())) return rc;
    int64_t offs[2] = {0, (int64_t)n};
    uint8_t* d_u8 = A.dev<uint8_t>(o_u8);
    int32_t* dn = A.dev<int32_t>(o_n);
    // EvImConverter::ev2im_gauss(l1Evs, W, H, sigma) :1345 (pol = false, normalized = true)
    // (the normalisation to u8 is left to the extraction's first kernel: one launch less)
    c->mm_preset = !zc;
    if ((rc = ev_accumulate_dev(c, A.in_ptr<void>(o_ev), is_raw, offs, 1, W, H, sigma, 0, 0, A.dev<float>(o_f32), d_u8, 0, A.dev<uint32_t>(o_mm)))) return rc;
    A.inputs_done();
    // makeFrame :1348 -> EvFrame ctor -> ORBextractor::operator() (EventFrame.cpp:220)
    c->pyr0_f32 = A.dev<float>(o_f32); c->pyr0_mm = A.dev<uint32_t>(o_mm);
    if ((rc = orb_extract_dev(c, d_u8, W, npix, 1, lap0, lap1, want_desc, A.dev<eorb_keypoint>(o_kp), A.dev<uint8_t>(o_desc), A.dev<uint8_t>(o_oob),
                              dn, dn + 1, dn + 2))) return rc;
    if ((rc = ensure(c, c->l1_ref_img, npix)) || (rc = ensure(c, c->l1_ref_pts, sizeof(float) * 2 * mo))) return rc;
    c->l1_nref = -1; c->l1_W = W; c->l1_H = H; c->klt_ref_serial++;
    c->l1_img_off = o_u8; c->l1_img_gen = c->arena_gen;
    const size_t ncopy = std::min<size_t>(mo, (size_t)std::max(cap, 0));
    const size_t first = out_u8 ? o_u8 : o_n;
    const size_t end = !ncopy ? o_n + 16 : (oob ? o_oob + ncopy : ((want_desc && desc) ? o_desc + 32 * ncopy : (kps ? o_kp + sizeof(eorb_keypoint) * ncopy : o_n + 16)));
    const char* h;
    if ((rc = A.download_begin(first, end - first))) return rc;
    // ELK_Tracker::setRefImage(image, keypoints) (:1363 init -> KLT_Tracker.cpp:22-46): the image and its points stay on the device --
    // queued behind the download, which does not wait for them (the next call on the stream is ordered behind them)
    EORB_HIP(c, hipMemcpyAsync(c->l1_ref_img.p, d_u8, npix, hipMemcpyDeviceToDevice, c->stream));
    if ((rc = ev_kp_points_dev(c, A.dev<eorb_keypoint>(o_kp), dn, (int)mo, (float*)c->l1_ref_pts.p))) return rc;
    if ((rc = A.download_wait(&h))) return rc;
    const int32_t* hn = (const int32_t*)(h + o_n);
    if (hn[2]) return set_err(c, EORB_E_CAPACITY, "ev_slice_extract: internal capacity exceeded (flag %d)", hn[2]);
    c->l1_nref = hn[0];
    if (out_u8) memcpy(out_u8, h + o_u8, npix);
    if (hn[0] > cap) return set_err(c, EORB_E_CAPACITY, "ev_slice_extract: %d keypoints > caller capacity %d", hn[0], cap);
    if (hn[0] > 0) {
        if (kps) memcpy(kps, h + o_kp, sizeof(eorb_keypoint) * (size_t)hn[0]);
        if (want_desc && desc) memcpy(desc, h + o_desc, 32 * (size_t)hn[0]);
        if (oob) memcpy(oob, h + o_oob, (size_t)hn[0]);
    }
    if (n_out) *n_out = hn[0];
    if (mono_index) *mono_index = hn[1];
    return EORB_OK;
}

int eorb_ev_slice_track(eorb_ctx* c, const eorb_event* ev, const eorb_raw_event* raw, size_t n, float sigma, const eorb_klt_params* klt,
                        float* pts, uint8_t* status, float* err, int nref, uint8_t* out_u8)
{
    if (!c) return EORB_E_ARG;
    if (!klt || !(sigma > 0.f) || nref < 0 || (nref && (!pts || !status || !err))) return set_err(c, EORB_E_ARG, "ev_slice_track: bad arguments");
    if (klt->win < 3 || klt->win > 63 || klt->maxLevel < 0) return set_err(c, EORB_E_ARG, "ev_slice_track: bad LK parameters");
    if (c->l1_nref < 0) return set_err(c, EORB_E_NOTCONF, "ev_slice_track: no reference frame (eorb_ev_slice_extract sets it)");
    if (nref != c->l1_nref) return set_err(c, EORB_E_ARG, "ev_slice_track: %d points, the reference frame has %d", nref, c->l1_nref);
    fe_enter(c);
    const int W = c->l1_W, H = c->l1_H;
    const size_t npix = (size_t)W * H;
    int rc, is_raw = 0;
    Arena A(c);
    std::vector<eorb_event16> packed;
    size_t o_ev = 0;
    if ((rc = slice_events_in(c, A, ev, raw, n, packed, &o_ev, &is_raw, "ev_slice_track"))) return rc;
    // in: the running extremes, initialised; in / out: the points (initial flow in, tracked points out); outputs behind them:
    // status | err | u8 image; then device-only
    const size_t o_mm = A.in(kMinMaxPreset, sizeof(kMinMaxPreset));
    const size_t o_pts = A.in(pts, sizeof(float) * 2 * (size_t)nref);
    const size_t o_st = A.reserve((size_t)nref + 16), o_err = A.reserve(sizeof(float) * (size_t)nref), o_u8 = A.reserve(npix);
    const size_t o_f32 = A.reserve(sizeof(float) * npix);
    if ((rc = A.upload())) return rc;
    int64_t offs[2] = {0, (int64_t)n};
    uint8_t* d_u8 = A.dev<uint8_t>(o_u8);
    c->mm_preset = true;
    if ((rc = ev_accumulate_dev(c, A.dev<void>(o_ev), is_raw, offs, 1, W, H, sigma, 0, 0, A.dev<float>(o_f32), d_u8, 1, A.dev<uint32_t>(o_mm)))) return rc;
    c->l1_img_off = o_u8; c->l1_img_gen = c->arena_gen;
    // ELK_Tracker::trackCurrImage (KLT_Tracker.cpp:49-74): calcOpticalFlowPyrLK(mRefFrame, currImage, mRefPoints, kpts, ..., OPTFLOW_USE_INITIAL_FLOW)
    if (nref && (rc = klt_track_dev(c, (const uint8_t*)c->l1_ref_img.p, d_u8, W, H, W, (const float*)c->l1_ref_pts.p, A.dev<float>(o_pts), nref, klt->win,
                                    klt->maxLevel, klt->maxCount, klt->epsilon, 4 /* OPTFLOW_USE_INITIAL_FLOW */, klt->minEigThreshold,
                                    A.dev<uint8_t>(o_st), A.dev<float>(o_err), c->klt_ref_serial))) return rc;
    const size_t end = out_u8 ? o_u8 + npix : o_err + sizeof(float) * (size_t)nref;
    const char* h;
    if ((rc = A.download(o_pts, end - o_pts, &h))) return rc;
    if (nref) {
        memcpy(pts, h + o_pts, sizeof(float) * 2 * (size_t)nref);
        memcpy(status, h + o_st, (size_t)nref);
        memcpy(err, h + o_err, sizeof(float) * (size_t)nref);
    }
    if (out_u8) memcpy(out_u8, h + o_u8, npix);
    return EORB_OK;
}

int eorb_ev_slice_image(eorb_ctx* c, uint8_t* out_u8)
{
    if (!c || !out_u8) return EORB_E_ARG;
    if (c->l1_img_gen != c->arena_gen || !c->arena.p || !c->l1_W) return set_err(c, EORB_E_NOTCONF, "ev_slice_image: the image of the last slice call is gone");
    fe_enter(c);
    EORB_HIP(c, hipMemcpyAsync(out_u8, (const char*)c->arena.p + c->l1_img_off, (size_t)c->l1_W * c->l1_H, hipMemcpyDeviceToHost, c->stream));
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

int eorb_ev_mc_contest(eorb_ctx* c, const eorb_event* ev, size_t n, const eorb_camera* cam, const eorb_se3_motion* dp, const eorb_se3_motion* ba,
                       const float* se2_params, int nparams, int W, int H, float sigma, float focus[5], int* winner, uint8_t* out_u8,
                       eorb_ctx* l2, int lap0, int lap1, eorb_keypoint* kps, int cap, int* n_out)
{
    if (!c) return EORB_E_ARG;
    if (winner) *winner = -1;
    if (n_out) *n_out = 0;
    if (W <= 0 || H <= 0 || (n && !ev) || !(sigma > 0.f) || !focus || !winner) return set_err(c, EORB_E_ARG, "ev_mc_contest: bad arguments");
    if ((dp || ba || se2_params) && !cam) return set_err(c, EORB_E_ARG, "ev_mc_contest: the motion-compensated methods need the camera");
    if (cam && cam->model != 0 && cam->model != 1) return set_err(c, EORB_E_ARG, "ev_mc_contest: camera model %d unknown", cam->model);
    if (se2_params && nparams < 3) return set_err(c, EORB_E_ARG, "ev_mc_contest: need at least 3 SE2 parameters");
    for (int k = 0; k < 5; k++) focus[k] = -1.f;
    if (n == 0) return EORB_OK;                          // "Empty ev buffer, abort" (:1149-1152)
    if (n > 0x3fffffff) return set_err(c, EORB_E_CAPACITY, "ev_mc_contest: too many events");
    if (l2) {
        if (l2->device != c->device) return set_err(c, EORB_E_ARG, "ev_mc_contest: the L2 context sits on another device");
        if (!l2->orb.configured || l2->orb.W != W || l2->orb.H != H) return set_err(c, EORB_E_NOTCONF, "ev_mc_contest: the L2 context's extractor is not configured for %dx%d", W, H);
    }
    fe_enter(c);
    const size_t npix = (size_t)W * H;
    int rc;
    // methods in the reference's insertion order (:1207-1211): 0 "DP", 1 "BA", 2 "EH", 3 "Opt"; image 4 = the event histogram of the
    // later half of the window (:1214-1216), built alongside so that the call waits once
    const eorb_se3_motion* se3[2] = {dp, ba};
    int img_of[4] = {-1, -1, -1, -1}, nimg = 0;
    for (int m = 0; m < 4; m++) if (m == 2 || (m < 2 && se3[m]) || (m == 3 && se2_params)) img_of[m] = nimg++;
    const int half_img = nimg++;
    const size_t nh = n / 2;                             // prefSize = evs.size() / 2: the window's last nh events (:1062-1066); 0 -> the whole window
    std::vector<eorb_event16> packed(n);
    eorb_pack_events(ev, n, packed.data());
    Arena A(c);
    const size_t o_ev = A.in(packed.data(), sizeof(eorb_event16) * n);
    const size_t o_warp = A.reserve(sizeof(eorb_event16) * n * 3);
    const size_t o_mm = A.reserve(256), o_f32 = A.reserve(sizeof(float) * npix * (size_t)nimg), o_u8s = A.reserve(npix * (size_t)nimg);
    const size_t o_fimg = A.reserve(64);
    const size_t mo = l2 ? (size_t)l2->orb.max_out : 0;
    // outputs, contiguous: focus[5] + winner | the winner's u8 image | {n, mono, flag} | keypoints
    const size_t o_res = A.reserve(64), o_win = A.reserve(npix), o_n = A.reserve(16), o_kp = A.reserve(sizeof(eorb_keypoint) * std::max<size_t>(mo, 1));
    if ((rc = A.upload())) return rc;
    const eorb_event16* d_ev = A.dev<eorb_event16>(o_ev);
    eorb_event16* d_warp = A.dev<eorb_event16>(o_warp);
    const int64_t wbase = (int64_t)((o_warp - o_ev) / sizeof(eorb_event16));      // (arena offsets are multiples of 256)
    int64_t beg[8], end[8];
    int nw = 0;
    for (int m = 0; m < 4; m++) {
        if (img_of[m] < 0) continue;
        const int j = img_of[m];
        if (m == 2) { beg[j] = 0; end[j] = (int64_t)n; continue; }                // getEvHist :1060-1079
        eorb_event16* dst = d_warp + (size_t)nw * n;
        if (m < 2) rc = ev_warp_se3_dev(c, d_ev, dst, (int)n, cam, se3[m]->angle, se3[m]->axis, se3[m]->t, se3[m]->medDepth, nullptr);       // getDPoseMCI :969 / getBAMCI :1043
        else rc = ev_warp_se2_dev(c, d_ev, dst, (int)n, cam, se2_params, nparams);                                                           // getAff2DMCI :1133
        if (rc) return rc;
        beg[j] = wbase + (int64_t)nw * (int64_t)n; end[j] = beg[j] + (int64_t)n;
        nw++;
    }
    beg[half_img] = nh ? (int64_t)(n - nh) : 0; end[half_img] = (int64_t)n;
    float* d_f32 = A.dev<float>(o_f32);
    uint32_t* d_mm = A.dev<uint32_t>(o_mm);
    static const long direct_max = [] { const char* e = getenv("EORB_CONTEST_DIRECT_MAX"); return e ? atol(e) : 49152L; }();      // (A/B runs)
    if ((long)n <= direct_max) {
        // every reconstruction of the window in ONE launch of the binning-free kernel (float events, normalized = false).  (Beyond
        // the single-slice limit of 16 384 events too: five binned passes of 60 us each are what the alternative costs here.)
        if ((rc = ev_direct_slices_dev(c, d_ev, 0, beg, end, nimg, W, H, sigma, 0, d_f32, nullptr, 0, d_mm))) return rc;
    } else {
        const int64_t dd_saved = c->dbg_dd_min; c->dbg_dd_min = 0;     // (warped events: no two share a position)
        for (int j = 0; j < nimg && !rc; j++) {
            int64_t offs[2] = {0, end[j] - beg[j]};
            rc = ev_accumulate_dev(c, d_ev + beg[j], 0, offs, 1, W, H, sigma, 0, 0, d_f32 + (size_t)j * npix, nullptr, 0, d_mm + 2 * j);
        }
        c->dbg_dd_min = dd_saved;
        if (rc) return rc;
    }
    // measureImageFocus of every image, then cv::normalize(img, img, 255, 0, NORM_MINMAX, CV_8UC1) of every image (:972-976, :1052-1055, :1073-1076, :1137-1140)
    if ((rc = ev_focus_dev(c, d_f32, nimg, W, H, A.dev<float>(o_fimg)))) return rc;
    if ((rc = ev_cvnormalize_n_dev(c, d_f32, nimg, (int)npix, d_mm + 32, A.dev<uint8_t>(o_u8s)))) return rc;
    float* d_res = A.dev<float>(o_res);
    if ((rc = ev_contest_select_dev(c, A.dev<float>(o_fimg), img_of, half_img, A.dev<uint8_t>(o_u8s), (int)npix, d_res, (int*)(d_res + 8), A.dev<uint8_t>(o_win)))) return rc;
    int32_t* dn = A.dev<int32_t>(o_n);
    if (l2) {
        // isMcImageGood (:260-267): the L2 tracker's makeFrame on the winner = its detect-only extraction.  The L2 context's kernels run on
        // THIS context's stream for the call (both contexts belong to the calling thread), so the call still waits once.
        if (l2 != c) EORB_HIP(c, hipStreamSynchronize(l2->stream));
        hipStream_t saved = l2->stream; l2->stream = c->stream;
        rc = orb_extract_dev(l2, A.dev<uint8_t>(o_win), W, npix, 1, lap0, lap1, 0, A.dev<eorb_keypoint>(o_kp), nullptr, nullptr, dn, dn + 1, dn + 2);
        l2->stream = saved;
        if (rc) { if (l2 != c) c->err = l2->err; return rc; }
    }
    const size_t ncopy = l2 ? std::min<size_t>(mo, (size_t)std::max(cap, 0)) : 0;
    const size_t end_off = l2 ? (ncopy && kps ? o_kp + sizeof(eorb_keypoint) * ncopy : o_n + 16) : (out_u8 ? o_win + npix : o_res + 64);
    const char* h;
    if ((rc = A.download(o_res, end_off - o_res, &h))) return rc;
    memcpy(focus, h + o_res, sizeof(float) * 5);
    int32_t w; memcpy(&w, h + o_res + 32, 4);
    *winner = w;
    if (out_u8) memcpy(out_u8, h + o_win, npix);
    if (l2) {
        const int32_t* hn = (const int32_t*)(h + o_n);
        if (hn[2]) return set_err(c, EORB_E_CAPACITY, "ev_mc_contest: the L2 extraction exceeded an internal capacity (flag %d)", hn[2]);
        if (hn[0] > cap) return set_err(c, EORB_E_CAPACITY, "ev_mc_contest: %d keypoints > caller capacity %d", hn[0], cap);
        if (hn[0] > 0 && kps) memcpy(kps, h + o_kp, sizeof(eorb_keypoint) * (size_t)hn[0]);
        if (n_out) *n_out = hn[0];
    }
    return EORB_OK;
}

int eorb_selfcheck_division(eorb_ctx* c, float lo, float hi, float sigma, uint64_t* mismatches)
{
    if (!c || !mismatches || !(lo > 0.f) || !(hi >= lo) || !(sigma > 0.f)) return c ? set_err(c, EORB_E_ARG, "selfcheck_division: bad arguments") : EORB_E_ARG;
    fe_enter(c);
    unsigned long long bad = 0;
    int rc = ev_divcheck(c, lo, hi, sigma, &bad);
    *mismatches = bad;
    return rc;
}

int eorb_selfcheck_math(eorb_ctx* c, int which, uint32_t lo_bits, uint32_t hi_bits, uint64_t* hash)
{
    if (!c || !hash || which < 0 || which > 5 || hi_bits < lo_bits) return c ? set_err(c, EORB_E_ARG, "selfcheck_math: bad arguments") : EORB_E_ARG;
    fe_enter(c);
    unsigned long long h = 0;
    int rc = ev_mathhash(c, which, lo_bits, hi_bits, &h);
    *hash = h;
    return rc;
}

#ifdef EORB_DIAG
int eorb_diag_read(unsigned long long* out16) { return ev_diag_read(out16); }
int eorb_trace_read(unsigned long long* out, int n) { return ev_trace_read(out, n); }
#endif

#ifdef EORB_SLOT_TRACE
int eorb_slot_trace_read(eorb_ctx* c, unsigned long long* out, long long max_records) { return c ? ev_slots_trace_read(c, out, max_records) : -1; }
#endif

// ---- ORB extractor, host buffers -----------------------------------------------------------------------
int eorb_orb_configure(eorb_ctx* c, const eorb_orb_params* p, int W, int H)
{
    if (!c) return EORB_E_ARG;
    fe_enter(c);
    hipStreamSynchronize(c->stream);
    return orb_configure(c, p, W, H);
}

int eorb_orb_max_keypoints(eorb_ctx* c) { return (c && c->orb.configured) ? c->orb.max_out : EORB_E_NOTCONF; }

int eorb_orb_get_tables(eorb_ctx* c, float* sf, float* inv_sf, int* nfeat, int* edge)
{
    if (!c || !c->orb.configured) return EORB_E_NOTCONF;
    for (int i = 0; i < c->orb.nlevels; i++) {
        if (sf) sf[i] = c->orb.sf[i];
        if (inv_sf) inv_sf[i] = c->orb.inv_sf[i];
        if (nfeat) nfeat[i] = c->orb.nfeat[i];
    }
    if (edge) *edge = c->orb.edge;
    return EORB_OK;
}

int eorb_orb_extract(eorb_ctx* c, const uint8_t* img, int W, int H, int stride, int lap0, int lap1,
                     int want_desc, eorb_keypoint* kps, uint8_t* desc, uint8_t* oob, int cap,
                     int* n_out, int* mono_index)
{
    if (!c) return EORB_E_ARG;
    if (n_out) *n_out = 0;
    if (!img || W <= 0 || H <= 0) return EORB_E_EMPTY;                 // _image.empty() -> -1 (:1096)
    OrbState& o = c->orb;
    if (!o.configured) return set_err(c, EORB_E_NOTCONF, "eorb_orb_extract: not configured");
    if (W != o.W || H != o.H) return set_err(c, EORB_E_ARG, "image %dx%d does not match the configured %dx%d", W, H, o.W, o.H);
    if (stride < W) return set_err(c, EORB_E_ARG, "stride < width");
    fe_enter(c);
    int rc;
    const size_t mo = (size_t)o.max_out;
    Arena A(c);
    // the image is read once, by the pyramid's first kernel: it reads the pinned staging buffer itself (no upload in front of the launch)
    static const int zc_env = [] { const char* e = getenv("EORB_IMAGE_ZERO_COPY"); return e ? atoi(e) : 1; }();      // (A/B runs)
    A.host_inputs = zc_env != 0 && (size_t)W * H <= ((size_t)1 << 20);
    const size_t o_img = A.in2d(img, H, (size_t)W, (size_t)stride);
    // outputs, contiguous: {n, mono, flag, pad} | keypoints | descriptors | oob
    const size_t o_n = A.reserve(16), o_kp = A.reserve(sizeof(eorb_keypoint) * mo), o_desc = A.reserve(32 * mo), o_oob = A.reserve(mo);
    if ((rc = A.upload())) return rc;
    int32_t* dn = A.dev<int32_t>(o_n);
    rc = orb_extract_dev(c, A.in_ptr<uint8_t>(o_img), W, (size_t)W * H, 1, lap0, lap1, want_desc, A.dev<eorb_keypoint>(o_kp),
                         A.dev<uint8_t>(o_desc), A.dev<uint8_t>(o_oob), dn, dn + 1, dn + 2);
    A.inputs_done();
    if (rc) return rc;
    // one copy back: counters always, the rest up to the caller's capacity
    const size_t ncopy = std::min<size_t>(mo, (size_t)std::max(cap, 0));
    const size_t end = !ncopy ? o_n + 16 : (oob ? o_oob + ncopy : ((want_desc && desc) ? o_desc + 32 * ncopy : (kps ? o_kp + sizeof(eorb_keypoint) * ncopy : o_n + 16)));
    const char* h;
    if ((rc = A.download(o_n, end - o_n, &h))) return rc;
    const int32_t* hn = (const int32_t*)(h + o_n);
    if (hn[2])                                    // (reported here; the sticky word of the *_dev calls is not involved)
        return set_err(c, EORB_E_CAPACITY, "orb_extract: internal capacity exceeded (flag %d)", hn[2]);
    if (hn[0] > cap) return set_err(c, EORB_E_CAPACITY, "orb_extract: %d keypoints > caller capacity %d", hn[0], cap);
    if (hn[0] > 0) {
        if (kps) memcpy(kps, h + o_kp, sizeof(eorb_keypoint) * (size_t)hn[0]);
        if (want_desc && desc) memcpy(desc, h + o_desc, 32 * (size_t)hn[0]);
        if (oob) memcpy(oob, h + o_oob, (size_t)hn[0]);
    }
    if (n_out) *n_out = hn[0];
    if (mono_index) *mono_index = hn[1];
    return EORB_OK;
}

// Frame::Frame(imLeft, imRight, ...) (src/Frame.cc:97-152): both images through the extractor (two slices of one batch: the
// reference's two extractors have equal parameters, :122-125 with vLappingArea {0, 0}), then ComputeStereoMatches (:869-1048)
int eorb_frame_stereo(eorb_ctx* c, const uint8_t* imLeft, const uint8_t* imRight, int W, int H, int stride, float mb, float mbf,
                      eorb_keypoint* kpsL, uint8_t* descL, int* nL, eorb_keypoint* kpsR, uint8_t* descR, int* nR, int cap,
                      float* uRight, float* depth, int* nmatches)
{
    if (!c) return EORB_E_ARG;
    if (nL) *nL = 0; if (nR) *nR = 0; if (nmatches) *nmatches = 0;
    if (!imLeft || !imRight || W <= 0 || H <= 0) return EORB_E_EMPTY;
    OrbState& o = c->orb;
    if (!o.configured) return set_err(c, EORB_E_NOTCONF, "eorb_frame_stereo: not configured");
    if (W != o.W || H != o.H || stride < W) return set_err(c, EORB_E_ARG, "eorb_frame_stereo: the images do not match the configured %dx%d", o.W, o.H);
    if (!(mb > 0.f) || !(mbf > 0.f)) return set_err(c, EORB_E_ARG, "eorb_frame_stereo: baseline %.4f, bf %.4f", mb, mbf);
    fe_enter(c);
    int rc;
    const size_t mo = (size_t)o.max_out;
    Arena A(c);
    const size_t o_imL = A.in2d(imLeft, H, (size_t)W, (size_t)stride), o_imR = A.in2d(imRight, H, (size_t)W, (size_t)stride);
    // outputs, contiguous: {n[2], mono[2], flag, matches, pad} | keypoints [2] | descriptors [2] | uRight | depth | (norms)
    const size_t o_n = A.reserve(32), o_kp = A.reserve(sizeof(eorb_keypoint) * mo * 2), o_desc = A.reserve(32 * mo * 2);
    const size_t o_ur = A.reserve(sizeof(float) * mo), o_dp = A.reserve(sizeof(float) * mo), o_sad = A.reserve(sizeof(int32_t) * mo);
    if ((rc = A.upload())) return rc;
    int32_t* dn = A.dev<int32_t>(o_n);
    rc = orb_extract_dev(c, A.dev<uint8_t>(o_imL), W, o_imR - o_imL, 2, 0, 0, 1, A.dev<eorb_keypoint>(o_kp), A.dev<uint8_t>(o_desc), nullptr, dn, dn + 2, dn + 4);
    if (rc) return rc;
    if ((rc = stereo_match_dev(c, A.dev<eorb_keypoint>(o_kp), A.dev<uint8_t>(o_desc), dn, mb, mbf, A.dev<float>(o_ur), A.dev<float>(o_dp),
                               A.dev<int32_t>(o_sad), dn + 5))) return rc;
    const char* h;
    if ((rc = A.download(o_n, o_sad - o_n, &h))) return rc;
    const int32_t* hn = (const int32_t*)(h + o_n);
    if (hn[4]) return set_err(c, EORB_E_CAPACITY, "eorb_frame_stereo: internal capacity exceeded (flag %d)", hn[4]);
    if (hn[0] > cap || hn[1] > cap) return set_err(c, EORB_E_CAPACITY, "eorb_frame_stereo: %d / %d keypoints > caller capacity %d", hn[0], hn[1], cap);
    if (hn[0] > 0) {
        if (kpsL) memcpy(kpsL, h + o_kp, sizeof(eorb_keypoint) * (size_t)hn[0]);
        if (descL) memcpy(descL, h + o_desc, 32 * (size_t)hn[0]);
        if (uRight) memcpy(uRight, h + o_ur, sizeof(float) * (size_t)hn[0]);
        if (depth) memcpy(depth, h + o_dp, sizeof(float) * (size_t)hn[0]);
    }
    if (hn[1] > 0) {
        if (kpsR) memcpy(kpsR, h + o_kp + sizeof(eorb_keypoint) * mo, sizeof(eorb_keypoint) * (size_t)hn[1]);
        if (descR) memcpy(descR, h + o_desc + 32 * mo, 32 * (size_t)hn[1]);
    }
    if (nL) *nL = hn[0];
    if (nR) *nR = hn[1];
    if (nmatches) *nmatches = hn[5];
    return EORB_OK;
}

static int tracked_common(eorb_ctx* c, const uint8_t* img, int W, int H, int stride, eorb_keypoint* kps_io, const eorb_keypoint* kps_in,
                          int n, int mode, const uint8_t* ref, uint8_t* desc, uint8_t* oob)
{
    if (!c) return EORB_E_ARG;
    if (!img || W <= 0 || H <= 0) return EORB_E_EMPTY;                 // trackedImage.empty() -> return (:1270, :1319)
    OrbState& o = c->orb;
    if (!o.configured) return set_err(c, EORB_E_NOTCONF, "tracked descriptors: not configured");
    if (W != o.W || H != o.H || stride < W || n < 0) return set_err(c, EORB_E_ARG, "tracked descriptors: bad image/arguments");
    if (n == 0) return EORB_OK;
    fe_enter(c);
    int rc;
    if ((rc = ensure(c, c->in_img, (size_t)W * H))) return rc;
    if ((rc = ensure(c, c->out_kp, sizeof(eorb_keypoint) * (size_t)std::max(n, o.max_out)))) return rc;
    if ((rc = ensure(c, c->m_a, 32 * (size_t)std::max(n, o.max_out)))) return rc;
    if ((rc = ensure(c, c->m_b, (size_t)std::max(n, o.max_out)))) return rc;
    if ((rc = ensure(c, c->m_c, 32 * (size_t)n))) return rc;
    if (stride == W) EORB_HIP(c, hipMemcpyAsync(c->in_img.p, img, (size_t)W * H, hipMemcpyHostToDevice, c->stream));
    else EORB_HIP(c, hipMemcpy2DAsync(c->in_img.p, W, img, stride, W, H, hipMemcpyHostToDevice, c->stream));
    EORB_HIP(c, hipMemcpyAsync(c->out_kp.p, kps_in, sizeof(eorb_keypoint) * n, hipMemcpyHostToDevice, c->stream));
    if (ref) EORB_HIP(c, hipMemcpyAsync(c->m_c.p, ref, 32 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if ((rc = orb_pyramid_blur_dev(c, (const uint8_t*)c->in_img.p, W))) return rc;
    if ((rc = orb_tracked_dev(c, (eorb_keypoint*)c->out_kp.p, n, mode, (const uint8_t*)c->m_c.p, (uint8_t*)c->m_a.p, (uint8_t*)c->m_b.p))) return rc;
    if (mode == 0) {
        EORB_HIP(c, hipMemcpyAsync(desc, c->m_a.p, 32 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        if (oob) EORB_HIP(c, hipMemcpyAsync(oob, c->m_b.p, n, hipMemcpyDeviceToHost, c->stream));
    } else {
        EORB_HIP(c, hipMemcpyAsync(kps_io, c->out_kp.p, sizeof(eorb_keypoint) * n, hipMemcpyDeviceToHost, c->stream));
    }
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

int eorb_orb_tracked_descriptors(eorb_ctx* c, const uint8_t* img, int W, int H, int stride, const eorb_keypoint* kps, int n,
                                 uint8_t* desc, uint8_t* oob)
{
    if (c && (!kps || !desc) && n > 0) return set_err(c, EORB_E_ARG, "tracked descriptors: null buffers");
    return tracked_common(c, img, W, H, stride, nullptr, kps, n, 0, nullptr, desc, oob);
}

int eorb_orb_assign_level_by_best_desc(eorb_ctx* c, const uint8_t* img, int W, int H, int stride, const uint8_t* ref_desc,
                                       eorb_keypoint* kps, int n)
{
    if (c && (!kps || !ref_desc) && n > 0) return set_err(c, EORB_E_ARG, "assign level: null buffers");
    return tracked_common(c, img, W, H, stride, kps, kps, n, 1, ref_desc, nullptr, nullptr);
}

// ---- matchers, host buffers ------------------------------------------------------------------------------
static int up_to(eorb_ctx* c, void* d_dst, const void* h, size_t bytes);
static int up(eorb_ctx* c, DevBuf& b, const void* h, size_t bytes)
{
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    return up_to(c, b.p, h, bytes);
}
// h -> d_dst (room for bytes rounded up to 16 there), staged in pinned memory and launched by up_flush()
static int up_to(eorb_ctx* c, void* d_dst, const void* h, size_t bytes)
{
    struct { void* p; } b{d_dst};
    if (!bytes || !h) return EORB_OK;
    static const long kmax = [] { const char* e = getenv("EORB_UPLOAD_KERNEL_MAX"); return e ? atol(e) : (1L << 20); }();
    void* p = (long)bytes <= kmax ? pin_bump(c->up_pin, bytes) : nullptr;
    if (!p) {                                           // (the caller's buffer may be a local: consumed before the return)
        EORB_HIP(c, hipMemcpyAsync(b.p, h, bytes, hipMemcpyHostToDevice, c->stream));
        EORB_HIP(c, hipStreamSynchronize(c->stream));
        return EORB_OK;
    }
    memcpy(p, h, bytes);
    c->up_queue.push_back({b.p, p, bytes});              // (launched by up_flush(): all of an entry's buffers in one kernel)
    return EORB_OK;
}

int eorb_search_for_initialization(eorb_ctx* c,
        const eorb_keypoint* kps1, int n1, const uint8_t* desc1, int stride1, const uint8_t* is_orb1,
        const eorb_keypoint* kps2, int n2, const uint8_t* desc2, int stride2, const uint8_t* is_orb2,
        const eorb_grid_bounds* gb, float* prev_matched, int32_t* matches12,
        int windowSize, float nnratio, int checkOri, int* nmatches)
{
    if (!c) return EORB_E_ARG;
    if (n1 < 0 || n2 < 0 || !gb || !matches12 || stride1 < 32 || stride2 < 32) return set_err(c, EORB_E_ARG, "search_for_initialization: bad arguments");
    fe_enter(c);
    if (nmatches) *nmatches = 0;
    if (n1 == 0) return EORB_OK;
    int rc;
    const int c1 = std::max(n1, 1), c2 = std::max(n2, 1);
    Arena A(c);
    int32_t hn[2] = {n1, n2};
    const size_t o_n = A.in(hn, 8);
    const size_t o_k1 = A.in(kps1, sizeof(eorb_keypoint) * (size_t)n1), o_d1 = A.in(desc1, (size_t)stride1 * n1);
    const size_t o_k2 = A.in(kps2, sizeof(eorb_keypoint) * (size_t)n2), o_d2 = A.in(desc2, (size_t)stride2 * n2);
    const size_t o_o1 = A.in(is_orb1, is_orb1 ? n1 : 0), o_o2 = A.in(is_orb2, is_orb2 ? n2 : 0);
    // outputs, contiguous: nmatches | matches12 | prev_matched (uploaded: it is in/out)
    const size_t o_nm = A.in(nullptr, 16);
    const size_t o_m = A.in(nullptr, sizeof(int32_t) * (size_t)c1);
    const size_t o_pm = A.in(prev_matched, prev_matched ? sizeof(float) * 2 * (size_t)n1 : 0);
    if ((rc = A.upload())) return rc;
    const int32_t* dn = A.dev<int32_t>(o_n);
    rc = search_init_dev(c, 1, A.dev<eorb_keypoint>(o_k1), dn, 0, A.dev<uint8_t>(o_d1), stride1, 0, is_orb1 ? A.dev<uint8_t>(o_o1) : nullptr,
                         A.dev<eorb_keypoint>(o_k2), dn + 1, 0, A.dev<uint8_t>(o_d2), stride2, 0, is_orb2 ? A.dev<uint8_t>(o_o2) : nullptr,
                         c1, c2, *gb, prev_matched ? A.dev<float>(o_pm) : nullptr, A.dev<int32_t>(o_m), windowSize, nnratio, checkOri,
                         A.dev<int32_t>(o_nm));
    if (rc) return rc;
    const size_t end = prev_matched ? o_pm + sizeof(float) * 2 * (size_t)n1 : o_m + sizeof(int32_t) * (size_t)n1;
    const char* h;
    if ((rc = A.download(o_nm, end - o_nm, &h))) return rc;
    memcpy(matches12, h + o_m, sizeof(int32_t) * (size_t)n1);
    if (prev_matched) memcpy(prev_matched, h + o_pm, sizeof(float) * 2 * (size_t)n1);
    if (nmatches) *nmatches = *(const int32_t*)(h + o_nm);
    return EORB_OK;
}

static int proj_last_common(eorb_ctx* c,
        const eorb_keypoint* cur_kps, int n_cur, const uint8_t* cur_desc, int cur_stride, const uint8_t* cur_is_orb,
        const eorb_keypoint* last_kps, int n_last, const uint8_t* last_is_orb,
        const uint8_t* valid, const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
        const float* level_scale, const eorb_grid_bounds* gb, int32_t* cur_mp, float th, int mode, int checkOri,
        int dist_th, int* nmatches, const float* cur_uright = nullptr, const float* q_ur = nullptr)
{
    if (!c) return EORB_E_ARG;
    if (n_cur < 0 || n_last < 0 || !gb || !cur_mp || cur_stride < 32 || !level_scale || ((cur_uright != nullptr) != (q_ur != nullptr)))
        return set_err(c, EORB_E_ARG, "search_by_projection_last: bad arguments");
    fe_enter(c);
    if (nmatches) *nmatches = 0;
    if (n_last == 0 || n_cur == 0) return EORB_OK;
    int rc;
    Arena A(c);
    const size_t o_ck = A.in(cur_kps, sizeof(eorb_keypoint) * (size_t)n_cur), o_cd = A.in(cur_desc, (size_t)cur_stride * n_cur);
    const size_t o_lk = A.in(last_kps, sizeof(eorb_keypoint) * (size_t)n_last), o_md = A.in(mp_desc, 32 * (size_t)n_last);
    const size_t o_co = A.in(cur_is_orb, cur_is_orb ? n_cur : 0), o_lo = A.in(last_is_orb, last_is_orb ? n_last : 0);
    std::vector<float> f3(3 * (size_t)n_last);
    for (int i = 0; i < n_last; i++) { f3[3 * i] = uv[2 * i]; f3[3 * i + 1] = uv[2 * i + 1]; f3[3 * i + 2] = level_scale[i]; }
    const size_t o_f3 = A.in(f3.data(), sizeof(float) * f3.size());
    const size_t o_va = A.in(valid, n_last), o_ob = A.in(mp_obs, n_last);
    const size_t o_ur2 = A.in(cur_uright, cur_uright ? sizeof(float) * (size_t)n_cur : 0), o_qur = A.in(q_ur, q_ur ? sizeof(float) * (size_t)n_last : 0);
    // outputs, contiguous: nmatches | slots (in/out)
    const size_t o_nm = A.in(nullptr, 16);
    const size_t o_mp = A.in(cur_mp, sizeof(int32_t) * (size_t)n_cur);
    if ((rc = A.upload())) return rc;
    rc = search_proj_last_dev(c, A.dev<eorb_keypoint>(o_ck), n_cur, A.dev<uint8_t>(o_cd), cur_stride,
                              cur_is_orb ? A.dev<uint8_t>(o_co) : nullptr, A.dev<eorb_keypoint>(o_lk), n_last,
                              last_is_orb ? A.dev<uint8_t>(o_lo) : nullptr, A.dev<uint8_t>(o_va), A.dev<float>(o_f3),
                              A.dev<uint8_t>(o_md), A.dev<uint8_t>(o_ob), dist_th, *gb, A.dev<int32_t>(o_mp), th,
                              mode, checkOri, A.dev<int32_t>(o_nm), cur_uright ? A.dev<float>(o_ur2) : nullptr, q_ur ? A.dev<float>(o_qur) : nullptr);
    if (rc) return rc;
    const char* h;
    if ((rc = A.download(o_nm, o_mp + sizeof(int32_t) * (size_t)n_cur - o_nm, &h))) return rc;
    memcpy(cur_mp, h + o_mp, sizeof(int32_t) * (size_t)n_cur);
    if (nmatches) *nmatches = *(const int32_t*)(h + o_nm);
    return EORB_OK;
}

int eorb_search_by_projection_last(eorb_ctx* c,
        const eorb_keypoint* cur_kps, int n_cur, const uint8_t* cur_desc, int cur_stride, const uint8_t* cur_is_orb,
        const eorb_keypoint* last_kps, int n_last, const uint8_t* last_is_orb,
        const uint8_t* valid, const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
        const float* level_scale, const eorb_grid_bounds* gb, int32_t* cur_mp, float th, int mode, int checkOri,
        int* nmatches)
{
    return proj_last_common(c, cur_kps, n_cur, cur_desc, cur_stride, cur_is_orb, last_kps, n_last, last_is_orb, valid, uv, mp_desc,
                            mp_obs, level_scale, gb, cur_mp, th, mode, checkOri, 100 /* TH_HIGH */, nmatches);
}

int eorb_search_by_projection_last_stereo(eorb_ctx* c,
        const eorb_keypoint* cur_kps, int n_cur, const uint8_t* cur_desc, int cur_stride, const uint8_t* cur_is_orb,
        const eorb_keypoint* last_kps, int n_last, const uint8_t* last_is_orb,
        const uint8_t* valid, const float* uv, const uint8_t* mp_desc, const uint8_t* mp_obs,
        const float* level_scale, const eorb_grid_bounds* gb, int32_t* cur_mp, float th, int mode, int checkOri,
        const float* cur_uright, const float* proj_ur, int* nmatches)
{
    if (c && (!cur_uright || !proj_ur)) return set_err(c, EORB_E_ARG, "search_by_projection_last_stereo: mvuRight and the projected right coordinates are needed");
    return proj_last_common(c, cur_kps, n_cur, cur_desc, cur_stride, cur_is_orb, last_kps, n_last, last_is_orb, valid, uv, mp_desc,
                            mp_obs, level_scale, gb, cur_mp, th, mode, checkOri, 100 /* TH_HIGH */, nmatches, cur_uright, proj_ur);
}

int eorb_search_by_projection_kf(eorb_ctx* c,
        const eorb_keypoint* cur_kps, int n_cur, const uint8_t* cur_desc, int cur_stride, const uint8_t* cur_is_orb,
        const eorb_keypoint* kf_kps, int n_kf, const uint8_t* kf_is_orb,
        const uint8_t* valid, const float* uv, const int32_t* pred_level, const float* level_scale, const uint8_t* mp_desc,
        const eorb_grid_bounds* gb, int32_t* cur_mp, float th, int ORBdist, int checkOri, int* nmatches)
{
    if (!c) return EORB_E_ARG;
    if (n_kf < 0 || (n_kf > 0 && (!kf_kps || !pred_level || !valid))) return set_err(c, EORB_E_ARG, "search_by_projection_kf: bad arguments");
    // the last-frame kernel with: query level = nPredictedLevel (:2236), window [L-1, L+1] (:2241), every occupied slot of the
    // current frame skipped (:2255-2256) and ORBdist in place of TH_HIGH (:2271)
    std::vector<eorb_keypoint> q(kf_kps, kf_kps + n_kf);
    for (int i = 0; i < n_kf; i++) { q[i].octave = pred_level[i]; q[i].class_id = pred_level[i]; }
    std::vector<uint8_t> obs((size_t)n_kf, 1);
    std::vector<int32_t> slots(cur_mp, cur_mp + (n_cur > 0 ? n_cur : 0));
    for (int i = 0; i < n_cur; i++) if (slots[i] != -1) slots[i] = -2;
    const int rc = proj_last_common(c, cur_kps, n_cur, cur_desc, cur_stride, cur_is_orb, q.data(), n_kf, kf_is_orb, valid, uv, mp_desc,
                                    obs.data(), level_scale, gb, slots.data(), th, 0, checkOri, ORBdist, nmatches);
    if (rc) return rc;
    for (int i = 0; i < n_cur; i++) if (slots[i] >= 0) cur_mp[i] = slots[i];
    return EORB_OK;
}

static int proj_map_common(eorb_ctx* c,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const uint8_t* is_orb,
        int M, const uint8_t* in_view, const float* proj_xy, const int32_t* level, const float* view_cos,
        const uint8_t* mp_desc, const uint8_t* mp_obs, const uint8_t* mp_is_orb, const float* level_scale,
        const eorb_grid_bounds* gb, int32_t* frame_mp, float th, float nnratio, int* nmatches, const float* uright, const float* proj_xr)
{
    if (!c) return EORB_E_ARG;
    if (n < 0 || M < 0 || !gb || !frame_mp || stride < 32 || ((uright != nullptr) != (proj_xr != nullptr))) return set_err(c, EORB_E_ARG, "search_by_projection_map: bad arguments");
    fe_enter(c);
    if (nmatches) *nmatches = 0;
    if (M == 0 || n == 0) return EORB_OK;
    int rc;
    Arena A(c);
    const size_t o_k = A.in(kps, sizeof(eorb_keypoint) * (size_t)n), o_d = A.in(desc, (size_t)stride * n), o_o = A.in(is_orb, is_orb ? n : 0);
    const size_t o_md = A.in(mp_desc, 32 * (size_t)M);
    // per map point record: proj x, proj y, view cos, level scale (floats) | level (int) | in_view, obs, is_orb (bytes)
    std::vector<float> f4(4 * (size_t)M);
    for (int m = 0; m < M; m++) { f4[4 * m] = proj_xy[2 * m]; f4[4 * m + 1] = proj_xy[2 * m + 1]; f4[4 * m + 2] = view_cos[m]; f4[4 * m + 3] = level_scale[m]; }
    const size_t o_f4 = A.in(f4.data(), sizeof(float) * f4.size()), o_lv = A.in(level, sizeof(int32_t) * (size_t)M);
    const size_t o_iv = A.in(in_view, M), o_ob = A.in(mp_obs, M), o_mo = A.in(mp_is_orb, mp_is_orb ? M : 0);
    const size_t o_ur2 = A.in(uright, uright ? sizeof(float) * (size_t)n : 0), o_qur = A.in(proj_xr, proj_xr ? sizeof(float) * (size_t)M : 0);
    const size_t o_nm = A.in(nullptr, 16);
    const size_t o_fm = A.in(frame_mp, sizeof(int32_t) * (size_t)n);
    if ((rc = A.upload())) return rc;
    rc = search_proj_map_dev(c, A.dev<eorb_keypoint>(o_k), n, A.dev<uint8_t>(o_d), stride, is_orb ? A.dev<uint8_t>(o_o) : nullptr, M,
                             A.dev<uint8_t>(o_iv), A.dev<float4>(o_f4), A.dev<int32_t>(o_lv), A.dev<uint8_t>(o_md), A.dev<uint8_t>(o_ob),
                             mp_is_orb ? A.dev<uint8_t>(o_mo) : nullptr, *gb, A.dev<int32_t>(o_fm), th, nnratio, A.dev<int32_t>(o_nm),
                             uright ? A.dev<float>(o_ur2) : nullptr, proj_xr ? A.dev<float>(o_qur) : nullptr);
    if (rc) return rc;
    const char* h;
    if ((rc = A.download(o_nm, o_fm + sizeof(int32_t) * (size_t)n - o_nm, &h))) return rc;
    memcpy(frame_mp, h + o_fm, sizeof(int32_t) * (size_t)n);
    if (nmatches) *nmatches = *(const int32_t*)(h + o_nm);
    return EORB_OK;
}

int eorb_search_by_projection_map(eorb_ctx* c,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const uint8_t* is_orb,
        int M, const uint8_t* in_view, const float* proj_xy, const int32_t* level, const float* view_cos,
        const uint8_t* mp_desc, const uint8_t* mp_obs, const uint8_t* mp_is_orb, const float* level_scale,
        const eorb_grid_bounds* gb, int32_t* frame_mp, float th, float nnratio, int* nmatches)
{
    return proj_map_common(c, kps, n, desc, stride, is_orb, M, in_view, proj_xy, level, view_cos, mp_desc, mp_obs, mp_is_orb, level_scale, gb,
                           frame_mp, th, nnratio, nmatches, nullptr, nullptr);
}

int eorb_search_by_projection_map_stereo(eorb_ctx* c,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const uint8_t* is_orb,
        int M, const uint8_t* in_view, const float* proj_xy, const int32_t* level, const float* view_cos,
        const uint8_t* mp_desc, const uint8_t* mp_obs, const uint8_t* mp_is_orb, const float* level_scale,
        const eorb_grid_bounds* gb, int32_t* frame_mp, float th, float nnratio, const float* uright, const float* proj_xr, int* nmatches)
{
    if (c && (!uright || !proj_xr)) return set_err(c, EORB_E_ARG, "search_by_projection_map_stereo: mvuRight and mTrackProjXR are needed");
    return proj_map_common(c, kps, n, desc, stride, is_orb, M, in_view, proj_xy, level, view_cos, mp_desc, mp_obs, mp_is_orb, level_scale, gb,
                           frame_mp, th, nnratio, nmatches, uright, proj_xr);
}

static int bow_common(eorb_ctx* c, int kf_kf,
        const eorb_keypoint* kf_kps, int n_kf, const uint8_t* kf_desc, const uint8_t* kf_has_mp,
        const uint32_t* kf_nodes, const int32_t* kf_node_off, const int32_t* kf_idx, int kf_nn,
        const eorb_keypoint* f_kps, int n_f, const uint8_t* f_desc, const uint8_t* f_has_mp,
        const uint32_t* f_nodes, const int32_t* f_node_off, const int32_t* f_idx, int f_nn,
        int32_t* match_out, float nnratio, int checkOri, int* nmatches)
{
    if (!c) return EORB_E_ARG;
    if (n_kf < 0 || n_f < 0 || kf_nn < 0 || f_nn < 0 || !match_out) return set_err(c, EORB_E_ARG, "search_by_bow: bad arguments");
    fe_enter(c);
    if (nmatches) *nmatches = 0;
    const int nout = kf_kf ? n_kf : n_f;
    for (int i = 0; i < nout; i++) match_out[i] = -1;
    if (n_kf == 0 || n_f == 0 || kf_nn == 0 || f_nn == 0) return EORB_OK;
    const int nki = kf_node_off[kf_nn], nfi = f_node_off[f_nn];
    for (int i = 0; i < nki; i++) if (kf_idx[i] < 0 || kf_idx[i] >= n_kf) return set_err(c, EORB_E_ARG, "search_by_bow: KeyFrame index out of range");
    for (int i = 0; i < nfi; i++) if (f_idx[i] < 0 || f_idx[i] >= n_f) return set_err(c, EORB_E_ARG, "search_by_bow: frame index out of range");
    int rc;
    if ((rc = up(c, c->m_a, kf_kps, sizeof(eorb_keypoint) * n_kf))) return rc;
    if ((rc = up(c, c->m_b, kf_desc, 32 * (size_t)n_kf))) return rc;
    if ((rc = up(c, c->m_c, f_kps, sizeof(eorb_keypoint) * n_f))) return rc;
    if ((rc = up(c, c->m_d, f_desc, 32 * (size_t)n_f))) return rc;
    std::vector<uint8_t> flags((size_t)n_kf + n_f, 1);
    memcpy(flags.data(), kf_has_mp, n_kf);
    if (f_has_mp) memcpy(flags.data() + n_kf, f_has_mp, n_f);
    if ((rc = up(c, c->m_e, flags.data(), flags.size()))) return rc;
    // CSR blocks: [kf_nodes | kf_off | kf_idx] and [f_nodes | f_off | f_idx]
    std::vector<int32_t> blk;
    blk.insert(blk.end(), (const int32_t*)kf_nodes, (const int32_t*)kf_nodes + kf_nn);
    blk.insert(blk.end(), kf_node_off, kf_node_off + kf_nn + 1);
    blk.insert(blk.end(), kf_idx, kf_idx + nki);
    const size_t fbase = blk.size();
    blk.insert(blk.end(), (const int32_t*)f_nodes, (const int32_t*)f_nodes + f_nn);
    blk.insert(blk.end(), f_node_off, f_node_off + f_nn + 1);
    blk.insert(blk.end(), f_idx, f_idx + nfi);
    if ((rc = up(c, c->m_f, blk.data(), sizeof(int32_t) * blk.size()))) return rc;
    if ((rc = ensure(c, c->m_h, sizeof(int32_t) * (size_t)(n_f + n_kf)))) return rc;
    if ((rc = ensure(c, c->m_g, (size_t)std::max(n_f, n_kf)))) return rc;
    if ((rc = ensure(c, c->m_j, sizeof(int32_t) * 40))) return rc;
    if ((rc = up_flush(c))) return rc;                  // (no wait here: up() has consumed the host buffers when it returns)
    const int32_t* B = (const int32_t*)c->m_f.p;
    int32_t* hist = (int32_t*)c->m_j.p;
    int32_t* d_match_f = (int32_t*)c->m_h.p;
    int32_t* d_match12 = d_match_f + n_f;
    rc = search_bow_dev(c, (const eorb_keypoint*)c->m_a.p, (const uint8_t*)c->m_b.p, (const uint8_t*)c->m_e.p,
                        (const uint32_t*)B, B + kf_nn, B + kf_nn + kf_nn + 1, kf_nn,
                        (const eorb_keypoint*)c->m_c.p, n_f, (const uint8_t*)c->m_d.p,
                        (const uint32_t*)(B + fbase), B + fbase + f_nn, B + fbase + f_nn + f_nn + 1, f_nn,
                        d_match_f, (int8_t*)c->m_g.p, hist, hist + 32, nnratio, checkOri, kf_kf,
                        (const uint8_t*)c->m_e.p + n_kf, d_match12, n_kf);
    if (rc) return rc;
    int nm = 0;
    { const int rc_dn = down(c, match_out, kf_kf ? d_match12 : d_match_f, sizeof(int32_t) * nout); if (rc_dn) return rc_dn; }
    { const int rc_dn = down(c, &nm, hist + 32, 4); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    if (nmatches) *nmatches = nm;
    return EORB_OK;
}

int eorb_search_by_bow(eorb_ctx* c,
        const eorb_keypoint* kf_kps, int n_kf, const uint8_t* kf_desc, const uint8_t* kf_has_mp,
        const uint32_t* kf_nodes, const int32_t* kf_node_off, const int32_t* kf_idx, int kf_nn,
        const eorb_keypoint* f_kps, int n_f, const uint8_t* f_desc,
        const uint32_t* f_nodes, const int32_t* f_node_off, const int32_t* f_idx, int f_nn,
        int32_t* match_f, float nnratio, int checkOri, int* nmatches)
{
    return bow_common(c, 0, kf_kps, n_kf, kf_desc, kf_has_mp, kf_nodes, kf_node_off, kf_idx, kf_nn, f_kps, n_f, f_desc, nullptr,
                      f_nodes, f_node_off, f_idx, f_nn, match_f, nnratio, checkOri, nmatches);
}

int eorb_search_by_bow_kf(eorb_ctx* c,
        const eorb_keypoint* kps1, int n1, const uint8_t* desc1, const uint8_t* has_mp1,
        const uint32_t* nodes1, const int32_t* node_off1, const int32_t* idx1, int nn1,
        const eorb_keypoint* kps2, int n2, const uint8_t* desc2, const uint8_t* has_mp2,
        const uint32_t* nodes2, const int32_t* node_off2, const int32_t* idx2, int nn2,
        int32_t* match12, float nnratio, int checkOri, int* nmatches)
{
    if (c && !has_mp2 && n2 > 0) return set_err(c, EORB_E_ARG, "search_by_bow_kf: has_mp2 is required");
    return bow_common(c, 1, kps1, n1, desc1, has_mp1, nodes1, node_off1, idx1, nn1, kps2, n2, desc2, has_mp2,
                      nodes2, node_off2, idx2, nn2, match12, nnratio, checkOri, nmatches);
}

int eorb_search_for_triangulation(eorb_ctx* c,
        const eorb_keypoint* kps1, int n1, const uint8_t* desc1, int stride1, const uint8_t* elig1,
        const uint32_t* nodes1, const int32_t* node_off1, const int32_t* idx1, int nn1,
        const eorb_keypoint* kps2, int n2, const uint8_t* desc2, int stride2, const uint8_t* elig2,
        const uint32_t* nodes2, const int32_t* node_off2, const int32_t* idx2, int nn2,
        const float* ep, const float* F12, const float* scale2, const float* sigma2_2, int nlevels,
        int bCoarse, int checkOri, int32_t* match12, int* nmatches)
{
    if (!c) return EORB_E_ARG;
    if (n1 < 0 || n2 < 0 || nn1 < 0 || nn2 < 0 || !match12 || stride1 < 32 || stride2 < 32 || !ep || !F12 || !scale2 || !sigma2_2 ||
        nlevels <= 0 || nlevels > 64)
        return set_err(c, EORB_E_ARG, "search_for_triangulation: bad arguments");
    fe_enter(c);
    if (nmatches) *nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0 || nn1 == 0 || nn2 == 0) return EORB_OK;
    const int nki = node_off1[nn1], nfi = node_off2[nn2];
    for (int i = 0; i < nki; i++) if (idx1[i] < 0 || idx1[i] >= n1) return set_err(c, EORB_E_ARG, "search_for_triangulation: pKF1 index out of range");
    for (int i = 0; i < nfi; i++) if (idx2[i] < 0 || idx2[i] >= n2) return set_err(c, EORB_E_ARG, "search_for_triangulation: pKF2 index out of range");
    for (int i = 0; i < n2; i++)
        if (elig2[i] && (kps2[i].octave < 0 || kps2[i].octave >= nlevels))
            return set_err(c, EORB_E_ARG, "search_for_triangulation: pKF2 keypoint %d has octave %d outside [0,%d)", i, kps2[i].octave, nlevels);
    int rc;
    if ((rc = up(c, c->m_a, kps1, sizeof(eorb_keypoint) * n1))) return rc;
    if ((rc = up(c, c->m_b, desc1, (size_t)stride1 * n1))) return rc;
    if ((rc = up(c, c->m_c, kps2, sizeof(eorb_keypoint) * n2))) return rc;
    if ((rc = up(c, c->m_d, desc2, (size_t)stride2 * n2))) return rc;
    std::vector<uint8_t> flags((size_t)n1 + n2);
    memcpy(flags.data(), elig1, n1); memcpy(flags.data() + n1, elig2, n2);
    if ((rc = up(c, c->m_e, flags.data(), flags.size()))) return rc;
    std::vector<int32_t> blk;
    blk.insert(blk.end(), (const int32_t*)nodes1, (const int32_t*)nodes1 + nn1);
    blk.insert(blk.end(), node_off1, node_off1 + nn1 + 1);
    blk.insert(blk.end(), idx1, idx1 + nki);
    const size_t fbase = blk.size();
    blk.insert(blk.end(), (const int32_t*)nodes2, (const int32_t*)nodes2 + nn2);
    blk.insert(blk.end(), node_off2, node_off2 + nn2 + 1);
    blk.insert(blk.end(), idx2, idx2 + nfi);
    if ((rc = up(c, c->m_f, blk.data(), sizeof(int32_t) * blk.size()))) return rc;
    std::vector<float> lv(2 * (size_t)nlevels);
    memcpy(lv.data(), scale2, sizeof(float) * nlevels); memcpy(lv.data() + nlevels, sigma2_2, sizeof(float) * nlevels);
    if ((rc = up(c, c->m_i, lv.data(), sizeof(float) * lv.size()))) return rc;
    if ((rc = ensure(c, c->m_h, sizeof(int32_t) * (size_t)n1))) return rc;
    if ((rc = ensure(c, c->m_g, (size_t)n1))) return rc;
    if ((rc = ensure(c, c->m_j, sizeof(int32_t) * 40))) return rc;
    if ((rc = up_flush(c))) return rc;                  // (no wait here: up() has consumed the host buffers when it returns)
    const int32_t* B = (const int32_t*)c->m_f.p;
    int32_t* hist = (int32_t*)c->m_j.p;
    TriArgs A{};
    A.kps1 = (const eorb_keypoint*)c->m_a.p; A.n1 = n1; A.desc1 = (const uint8_t*)c->m_b.p; A.stride1 = stride1;
    A.elig1 = (const uint8_t*)c->m_e.p;
    A.nodes1 = (const uint32_t*)B; A.off1 = B + nn1; A.idx1 = B + nn1 + nn1 + 1; A.nn1 = nn1;
    A.kps2 = (const eorb_keypoint*)c->m_c.p; A.n2 = n2; A.desc2 = (const uint8_t*)c->m_d.p; A.stride2 = stride2;
    A.elig2 = (const uint8_t*)c->m_e.p + n1;
    A.nodes2 = (const uint32_t*)(B + fbase); A.off2 = B + fbase + nn2; A.idx2 = B + fbase + nn2 + nn2 + 1; A.nn2 = nn2;
    A.epx = ep[0]; A.epy = ep[1];
    for (int i = 0; i < 9; i++) A.F[i] = F12[i];
    A.scale2 = (const float*)c->m_i.p; A.sigma2_2 = (const float*)c->m_i.p + nlevels; A.nlevels = nlevels;
    A.bCoarse = bCoarse; A.checkOri = checkOri;
    A.match12 = (int32_t*)c->m_h.p; A.bin1 = (int8_t*)c->m_g.p; A.histo = hist; A.nmatches = hist + 32;
    if ((rc = search_tri_dev(c, A))) return rc;
    int nm = 0;
    { const int rc_dn = down(c, match12, c->m_h.p, sizeof(int32_t) * (size_t)n1); if (rc_dn) return rc_dn; }
    { const int rc_dn = down(c, &nm, hist + 32, 4); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    if (nmatches) *nmatches = nm;
    return EORB_OK;
}

static int kf_radius_common(eorb_ctx* c,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const eorb_grid_bounds* gb,
        int M, const uint8_t* valid, const float* uv, const float* radius, const int32_t* level, const uint8_t* q_desc,
        const float* inv_sigma2, int nlevels, uint8_t* taken, float accept_thr, int32_t* best_idx, int32_t* best_dist,
        const float* uright, const float* q_ur)
{
    if (!c) return EORB_E_ARG;
    if ((uright != nullptr) != (q_ur != nullptr) || (uright && !inv_sigma2)) return set_err(c, EORB_E_ARG, "kf_radius_match: the stereo gate needs uright, q_ur and inv_sigma2");
    if (n < 0 || M < 0 || stride < 32 || !gb || (M > 0 && (!valid || !uv || !radius || !level || !q_desc || !best_idx || !best_dist)) ||
        (inv_sigma2 && (nlevels <= 0 || nlevels > 64)))
        return set_err(c, EORB_E_ARG, "kf_radius_match: bad arguments");
    fe_enter(c);
    for (int m = 0; m < M; m++) { best_idx[m] = -1; best_dist[m] = 256; }
    if (M == 0 || n == 0) return EORB_OK;
    int rc;
    if ((rc = up(c, c->m_a, kps, sizeof(eorb_keypoint) * n))) return rc;
    if ((rc = up(c, c->m_b, desc, (size_t)stride * n))) return rc;
    if ((rc = up(c, c->m_c, uv, sizeof(float) * 2 * (size_t)M))) return rc;
    std::vector<int32_t> qi(2 * (size_t)M);
    memcpy(qi.data(), radius, sizeof(float) * M); memcpy(qi.data() + M, level, sizeof(int32_t) * M);
    if ((rc = up(c, c->m_d, qi.data(), sizeof(int32_t) * qi.size()))) return rc;
    std::vector<uint8_t> fl((size_t)M + n, 0);
    memcpy(fl.data(), valid, M);
    if (taken) memcpy(fl.data() + M, taken, n);
    if ((rc = up(c, c->m_e, fl.data(), fl.size()))) return rc;
    if ((rc = up(c, c->m_f, q_desc, 32 * (size_t)M))) return rc;
    if (inv_sigma2 && (rc = up(c, c->m_i, inv_sigma2, sizeof(float) * nlevels))) return rc;
    if (uright) {                                    // uright[n] | q_ur[M]
        std::vector<float> st((size_t)n + M);
        memcpy(st.data(), uright, sizeof(float) * n); memcpy(st.data() + n, q_ur, sizeof(float) * M);
        if ((rc = up(c, c->m_j, st.data(), sizeof(float) * st.size()))) return rc;
    }
    if ((rc = ensure(c, c->m_h, sizeof(int32_t) * 2 * (size_t)M))) return rc;
    if ((rc = ensure(c, c->m_g, sizeof(uint16_t) * (size_t)n))) return rc;
    if ((rc = up_flush(c))) return rc;
    RadArgs A{};
    A.kps = (const eorb_keypoint*)c->m_a.p; A.n = n; A.desc = (const uint8_t*)c->m_b.p; A.stride = stride;
    A.g = GridB{gb->minX, gb->minY, gb->invW, gb->invH};
    A.cell = (const uint16_t*)c->m_g.p;
    A.M = M; A.valid = (const uint8_t*)c->m_e.p; A.uv = (const float*)c->m_c.p;
    A.radius = (const float*)c->m_d.p; A.level = (const int32_t*)c->m_d.p + M; A.q_desc = (const uint8_t*)c->m_f.p;
    A.inv_sigma2 = inv_sigma2 ? (const float*)c->m_i.p : nullptr; A.nlevels = nlevels;
    A.uright = uright ? (const float*)c->m_j.p : nullptr; A.q_ur = uright ? (const float*)c->m_j.p + n : nullptr;
    A.taken = taken ? (uint8_t*)c->m_e.p + M : nullptr; A.accept_thr = accept_thr;
    A.best_idx = (int32_t*)c->m_h.p; A.best_dist = (int32_t*)c->m_h.p + M;
    if ((rc = kf_radius_dev(c, A, (uint16_t*)c->m_g.p))) return rc;
    { const int rc_dn = down(c, best_idx, c->m_h.p, sizeof(int32_t) * (size_t)M); if (rc_dn) return rc_dn; }
    { const int rc_dn = down(c, best_dist, (int32_t*)c->m_h.p + M, sizeof(int32_t) * (size_t)M); if (rc_dn) return rc_dn; }
    if (taken) { const int rc_dn = down(c, taken, (uint8_t*)c->m_e.p + M, (size_t)n); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

int eorb_kf_radius_match(eorb_ctx* c,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const eorb_grid_bounds* gb,
        int M, const uint8_t* valid, const float* uv, const float* radius, const int32_t* level, const uint8_t* q_desc,
        const float* inv_sigma2, int nlevels, uint8_t* taken, float accept_thr, int32_t* best_idx, int32_t* best_dist)
{
    return kf_radius_common(c, kps, n, desc, stride, gb, M, valid, uv, radius, level, q_desc, inv_sigma2, nlevels, taken, accept_thr, best_idx, best_dist, nullptr, nullptr);
}

int eorb_kf_radius_match_stereo(eorb_ctx* c,
        const eorb_keypoint* kps, int n, const uint8_t* desc, int stride, const eorb_grid_bounds* gb,
        int M, const uint8_t* valid, const float* uv, const float* radius, const int32_t* level, const uint8_t* q_desc,
        const float* inv_sigma2, int nlevels, const float* uright, const float* q_ur, int32_t* best_idx, int32_t* best_dist)
{
    return kf_radius_common(c, kps, n, desc, stride, gb, M, valid, uv, radius, level, q_desc, inv_sigma2, nlevels, nullptr, 0.f, best_idx, best_dist, uright, q_ur);
}

int eorb_bow_set_vocabulary(eorb_ctx* c, int nnodes, int L, const int32_t* child_off, const int32_t* child_ids,
                            const uint8_t* node_desc, const int32_t* word_id, const double* weight)
{
    if (!c) return EORB_E_ARG;
    if (nnodes < 1 || L < 1 || L > 32 || !child_off || !child_ids || !node_desc || !word_id || !weight)
        return set_err(c, EORB_E_ARG, "bow_set_vocabulary: bad arguments");
    // a tree rooted at node 0: monotone offsets, every node but the root has exactly one parent, depth <= 32, no cycles
    const int nch = child_off[nnodes];
    if (child_off[0] != 0 || nch != nnodes - 1) return set_err(c, EORB_E_ARG, "bow_set_vocabulary: %d child links for %d nodes", nch, nnodes);
    std::vector<int8_t> depth(nnodes, -1);
    depth[0] = 0;
    std::vector<int> stack{0};
    int visited = 0;
    while (!stack.empty()) {
        const int u = stack.back(); stack.pop_back(); visited++;
        if (child_off[u + 1] < child_off[u]) return set_err(c, EORB_E_ARG, "bow_set_vocabulary: offsets not monotone at node %d", u);
        for (int k = child_off[u]; k < child_off[u + 1]; k++) {
            const int v = child_ids[k];
            if (v <= 0 || v >= nnodes || depth[v] >= 0) return set_err(c, EORB_E_ARG, "bow_set_vocabulary: node %d is not a tree child", v);
            if (depth[u] >= 32) return set_err(c, EORB_E_ARG, "bow_set_vocabulary: tree deeper than 32");
            depth[v] = (int8_t)(depth[u] + 1);
            stack.push_back(v);
        }
    }
    if (visited != nnodes) return set_err(c, EORB_E_ARG, "bow_set_vocabulary: %d of %d nodes reachable from the root", visited, nnodes);
    fe_enter(c);
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t off[6]; off[0] = 0;
    off[1] = off[0] + al(sizeof(int32_t) * ((size_t)nnodes + 1));
    off[2] = off[1] + al(sizeof(int32_t) * (size_t)std::max(nch, 1));
    off[3] = off[2] + al(32 * (size_t)nnodes);
    off[4] = off[3] + al(sizeof(int32_t) * (size_t)nnodes);
    off[5] = off[4] + al(sizeof(double) * (size_t)nnodes);
    std::vector<uint8_t> blob(off[5], 0);
    memcpy(blob.data() + off[0], child_off, sizeof(int32_t) * ((size_t)nnodes + 1));
    memcpy(blob.data() + off[1], child_ids, sizeof(int32_t) * (size_t)nch);
    memcpy(blob.data() + off[2], node_desc, 32 * (size_t)nnodes);
    memcpy(blob.data() + off[3], word_id, sizeof(int32_t) * (size_t)nnodes);
    memcpy(blob.data() + off[4], weight, sizeof(double) * (size_t)nnodes);
    int rc;
    if ((rc = up(c, c->voc, blob.data(), blob.size()))) return rc;
    if ((rc = up_flush(c))) return rc;
    EORB_HIP(c, fe_stream_sync(c));
    c->voc_nnodes = nnodes; c->voc_L = L;
    for (int i = 0; i < 5; i++) c->voc_off[i] = off[i];
    return EORB_OK;
}

int eorb_bow_transform(eorb_ctx* c, const uint8_t* desc, int n, int stride, int levelsup, int weighting, int norm,
                       uint32_t* bow_word, double* bow_val, int* n_words, uint32_t* fv_node, int32_t* fv_off, int32_t* fv_idx,
                       int* n_fvnodes, int32_t* word_of, int32_t* node_of)
{
    if (!c) return EORB_E_ARG;
    if (!c->voc_nnodes) return set_err(c, EORB_E_NOTCONF, "bow_transform: eorb_bow_set_vocabulary not called");
    if (n < 0 || stride < 32 || weighting < 0 || weighting > 3 || norm < 0 || norm > 2 || !n_words || !n_fvnodes || !fv_off ||
        (n > 0 && (!desc || !bow_word || !bow_val || !fv_node || !fv_idx)))
        return set_err(c, EORB_E_ARG, "bow_transform: bad arguments");
    fe_enter(c);
    *n_words = 0; *n_fvnodes = 0; fv_off[0] = 0;
    if (n == 0 || c->voc_nnodes <= 1) return EORB_OK;                       // empty() (:1132)
    int rc;
    if ((rc = up(c, c->m_a, desc, (size_t)stride * n))) return rc;
    if ((rc = up_flush(c))) return rc;
    // workspace: word_of u32 | node_of u32 | bow_word u32 | fv_node u32 | fv_off i32 (+1) | fv_idx i32 | counts | w_of f64 | bow_val f64
    const size_t N = (size_t)n;
    if ((rc = ensure(c, c->m_b, 4 * (6 * N + 8) + 8 * (2 * N + 2)))) return rc;
    uint32_t* w32 = (uint32_t*)c->m_b.p;
    uint32_t* d_word_of = w32, *d_node_of = w32 + N, *d_bow_word = w32 + 2 * N, *d_fv_node = w32 + 3 * N;
    int32_t* d_fv_off = (int32_t*)(w32 + 4 * N), *d_fv_idx = (int32_t*)(w32 + 5 * N + 2), *d_counts = (int32_t*)(w32 + 6 * N + 4);
    double* d_w_of = (double*)(w32 + 6 * N + 8), *d_bow_val = d_w_of + N + 1;
    const char* vb = (const char*)c->voc.p;
    BowVoc V{c->voc_nnodes, c->voc_L, (const int32_t*)(vb + c->voc_off[0]), (const int32_t*)(vb + c->voc_off[1]),
             (const uint8_t*)(vb + c->voc_off[2]), (const int32_t*)(vb + c->voc_off[3]), (const double*)(vb + c->voc_off[4])};
    if ((rc = bow_transform_dev(c, (const uint8_t*)c->m_a.p, n, stride, V, levelsup, weighting, norm, d_word_of, d_w_of, d_node_of,
                                d_bow_word, d_bow_val, d_fv_node, d_fv_off, d_fv_idx, d_counts))) return rc;
    int32_t cnt[2] = {0, 0};
    { const int rc_dn = down(c, cnt, d_counts, 8); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    *n_words = cnt[0]; *n_fvnodes = cnt[1];
    if (cnt[0]) {
        { const int rc_dn = down(c, bow_word, d_bow_word, 4 * (size_t)cnt[0]); if (rc_dn) return rc_dn; }
        { const int rc_dn = down(c, bow_val, d_bow_val, 8 * (size_t)cnt[0]); if (rc_dn) return rc_dn; }
    }
    { const int rc_dn = down(c, fv_off, d_fv_off, 4 * ((size_t)cnt[1] + 1)); if (rc_dn) return rc_dn; }
    if (cnt[1]) { const int rc_dn = down(c, fv_node, d_fv_node, 4 * (size_t)cnt[1]); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    const int nfeat = fv_off[cnt[1]];
    if (nfeat) { const int rc_dn = down(c, fv_idx, d_fv_idx, 4 * (size_t)nfeat); if (rc_dn) return rc_dn; }
    if (word_of) { const int rc_dn = down(c, word_of, d_word_of, 4 * N); if (rc_dn) return rc_dn; }
    if (node_of) { const int rc_dn = down(c, node_of, d_node_of, 4 * N); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

int eorb_calc_optical_flow_pyr_lk(eorb_ctx* c, const uint8_t* prev, const uint8_t* next, int W, int H, int stride,
                                  const float* prev_pts, float* next_pts, int n, int win, int maxLevel, int maxCount, double epsilon,
                                  int flags, float minEigThreshold, uint8_t* status, float* err)
{
    if (!c) return EORB_E_ARG;
    if (!prev || !next || W <= 0 || H <= 0 || stride < W || n < 0 || win < 3 || win > 63 || maxLevel < 0 ||
        (n > 0 && (!prev_pts || !next_pts || !status || !err)))
        return set_err(c, EORB_E_ARG, "calc_optical_flow_pyr_lk: bad arguments");
    fe_enter(c);
    if (n == 0) return EORB_OK;
    int rc;
    const size_t ib = (size_t)stride * H;
    if ((rc = ensure(c, c->in_img, 2 * ib))) return rc;
    if ((rc = up_to(c, c->in_img.p, prev, ib)) || (rc = up_to(c, (uint8_t*)c->in_img.p + ib, next, ib))) return rc;
    if ((rc = up(c, c->m_a, prev_pts, sizeof(float) * 2 * (size_t)n))) return rc;
    if ((rc = up(c, c->m_b, next_pts, sizeof(float) * 2 * (size_t)n))) return rc;
    if ((rc = up_flush(c))) return rc;
    if ((rc = ensure(c, c->m_c, (size_t)n + 16))) return rc;
    if ((rc = ensure(c, c->m_d, sizeof(float) * (size_t)n))) return rc;
    if ((rc = klt_track_dev(c, (const uint8_t*)c->in_img.p, (const uint8_t*)c->in_img.p + ib, W, H, stride, (const float*)c->m_a.p,
                            (float*)c->m_b.p, n, win, maxLevel, maxCount, epsilon, flags, minEigThreshold, (uint8_t*)c->m_c.p,
                            (float*)c->m_d.p))) return rc;
    { const int rc_dn = down(c, next_pts, c->m_b.p, sizeof(float) * 2 * (size_t)n); if (rc_dn) return rc_dn; }
    { const int rc_dn = down(c, status, c->m_c.p, (size_t)n); if (rc_dn) return rc_dn; }
    { const int rc_dn = down(c, err, c->m_d.p, sizeof(float) * (size_t)n); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

int eorb_hamming_window_match(eorb_ctx* c, const uint8_t* q_desc, int nq, int q_stride, const uint8_t* t_desc, int nt, int t_stride,
                              const int32_t* cand_offsets, const int32_t* cand_idx, int32_t* best_idx, int32_t* best_d,
                              int32_t* second_idx, int32_t* second_d)
{
    if (!c) return EORB_E_ARG;
    if (nq < 0 || nt < 0 || q_stride < 32 || t_stride < 32 || (nq > 0 && (!q_desc || !cand_offsets || !best_idx || !best_d || !second_idx || !second_d)))
        return set_err(c, EORB_E_ARG, "hamming_window_match: bad arguments");
    fe_enter(c);
    if (nq == 0) return EORB_OK;
    const int ncand = cand_offsets[nq];
    for (int q = 0; q < nq; q++) if (cand_offsets[q + 1] < cand_offsets[q]) return set_err(c, EORB_E_ARG, "hamming_window_match: offsets not monotone");
    for (int k = 0; k < ncand; k++) if (cand_idx[k] < 0 || cand_idx[k] >= nt) return set_err(c, EORB_E_ARG, "hamming_window_match: candidate %d out of range", cand_idx[k]);
    int rc;
    if ((rc = up(c, c->m_a, q_desc, (size_t)q_stride * nq))) return rc;
    if ((rc = up(c, c->m_b, t_desc, (size_t)t_stride * std::max(nt, 1)))) return rc;
    if ((rc = up(c, c->m_c, cand_offsets, sizeof(int32_t) * ((size_t)nq + 1)))) return rc;
    if ((rc = up(c, c->m_d, cand_idx, sizeof(int32_t) * (size_t)std::max(ncand, 1)))) return rc;
    if ((rc = up_flush(c))) return rc;
    if ((rc = ensure(c, c->m_h, sizeof(int32_t) * 4 * (size_t)nq))) return rc;
    if ((rc = window_match_dev(c, (const uint8_t*)c->m_a.p, nq, q_stride, (const uint8_t*)c->m_b.p, t_stride, (const int32_t*)c->m_c.p,
                               (const int32_t*)c->m_d.p, (int32_t*)c->m_h.p))) return rc;
    int32_t* o = (int32_t*)c->m_h.p;
    { const int rc_dn = down(c, best_idx, o, 4 * (size_t)nq); if (rc_dn) return rc_dn; }
    { const int rc_dn = down(c, best_d, o + nq, 4 * (size_t)nq); if (rc_dn) return rc_dn; }
    { const int rc_dn = down(c, second_idx, o + 2 * (size_t)nq, 4 * (size_t)nq); if (rc_dn) return rc_dn; }
    { const int rc_dn = down(c, second_d, o + 3 * (size_t)nq, 4 * (size_t)nq); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

int eorb_distinctive_descriptors(eorb_ctx* c, const uint8_t* desc, const int32_t* offsets, int M, int32_t* best)
{
    if (!c) return EORB_E_ARG;
    if (M < 0 || (M > 0 && (!offsets || !best))) return set_err(c, EORB_E_ARG, "distinctive_descriptors: bad arguments");
    if (M == 0) return EORB_OK;
    for (int m = 0; m < M; m++) if (offsets[m + 1] < offsets[m]) return set_err(c, EORB_E_ARG, "distinctive_descriptors: offsets not monotone");
    fe_enter(c);
    const int n = offsets[M];
    int rc;
    if ((rc = up(c, c->m_a, desc, 32 * (size_t)std::max(n, 1)))) return rc;
    if ((rc = up(c, c->m_b, offsets, sizeof(int32_t) * (size_t)(M + 1)))) return rc;
    if ((rc = up_flush(c))) return rc;
    if ((rc = ensure(c, c->m_h, sizeof(int32_t) * (size_t)M))) return rc;
    if ((rc = distinctive_dev(c, (const uint8_t*)c->m_a.p, (const int32_t*)c->m_b.p, M, (int32_t*)c->m_h.p))) return rc;
    { const int rc_dn = down(c, best, c->m_h.p, sizeof(int32_t) * (size_t)M); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

int eorb_sort_by_response(eorb_ctx* c, const eorb_keypoint* kps, int n, int32_t* perm)
{
    if (!c) return EORB_E_ARG;
    if (n < 0 || (n > 0 && (!kps || !perm))) return set_err(c, EORB_E_ARG, "sort_by_response: bad arguments");
    if (n == 0) return EORB_OK;
    fe_enter(c);
    int rc;
    if ((rc = up(c, c->m_a, kps, sizeof(eorb_keypoint) * n))) return rc;
    if ((rc = up_flush(c))) return rc;
    if ((rc = ensure(c, c->m_h, sizeof(int32_t) * n))) return rc;
    if ((rc = sort_response_dev(c, (const eorb_keypoint*)c->m_a.p, n, (int32_t*)c->m_h.p))) return rc;
    { const int rc_dn = down(c, perm, c->m_h.p, sizeof(int32_t) * n); if (rc_dn) return rc_dn; }
    EORB_HIP(c, fe_stream_sync(c));
    return EORB_OK;
}

void eorb_resolve_num_mixed(int nDetectedORB, int nDetectedAK, int nDesired, int nDesiredAK, int* nORB, int* nAK)
{   // MixedFrame::resolveNumMixedPts (src/MixedFrame.cpp:281-317): pure count bookkeeping of the container
    const int nDetected = nDetectedORB + nDetectedAK;
    const int nDesiredORB = nDesired - nDesiredAK;
    if (nDetected > nDesired) {
        const int nDiff = nDetected - nDesired;
        if (nDetectedORB > nDesiredORB && nDetectedAK > nDesiredAK) { *nORB = nDesiredORB; *nAK = nDesiredAK; }
        else if (nDetectedORB > nDesiredORB) { *nORB = nDetectedORB - nDiff; *nAK = std::min(nDetectedAK, nDesiredAK); }
        else if (nDetectedAK > nDesiredAK) { *nORB = std::min(nDetectedORB, nDesiredORB); *nAK = nDetectedAK - nDiff; }
    } else { *nORB = nDetectedORB; *nAK = nDetectedAK; }
}

int eorb_hamming_bf_knn2(eorb_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2, int32_t* dist2)
{
    if (!c) return EORB_E_ARG;
    if (nq < 0 || nt < 0 || !idx2 || !dist2) return set_err(c, EORB_E_ARG, "bf_knn2: bad arguments");
    if (nq == 0) return EORB_OK;
    fe_enter(c);
    int rc;
    Arena A(c);
    const size_t o_q = A.in(q, 32 * (size_t)nq), o_t = A.in(t, 32 * (size_t)nt);
    const size_t o_idx = A.reserve(sizeof(int32_t) * 2 * (size_t)nq), o_dist = A.reserve(sizeof(int32_t) * 2 * (size_t)nq);
    if ((rc = A.upload())) return rc;
    rc = bf_knn2_dev(c, A.dev<uint8_t>(o_q), nq, A.dev<uint8_t>(o_t), nt, A.dev<int32_t>(o_idx), A.dev<int32_t>(o_dist));
    if (rc) return rc;
    const char* h;
    if ((rc = A.download(o_idx, o_dist + sizeof(int32_t) * 2 * (size_t)nq - o_idx, &h))) return rc;
    memcpy(idx2, h + o_idx, sizeof(int32_t) * 2 * (size_t)nq);
    memcpy(dist2, h + o_dist, sizeof(int32_t) * 2 * (size_t)nq);
    return EORB_OK;
}

// ---- batched HBM-resident front end --------------------------------------------------------------------
int eorb_fe_configure(eorb_ctx* c, const eorb_fe_config* cfg)
{
    if (!c || !cfg) return EORB_E_ARG;
    if (cfg->W <= 0 || cfg->H <= 0 || cfg->max_batch < 1 || cfg->max_events < 0 || !(cfg->sigma > 0.f))
        return set_err(c, EORB_E_ARG, "fe_configure: bad configuration");
    fe_enter(c);
    hipStreamSynchronize(c->stream);
    int rc = orb_configure(c, &cfg->orb, cfg->W, cfg->H);
    if (rc) return rc;
    c->fe = *cfg;
    const size_t B = cfg->max_batch, npix = (size_t)cfg->W * cfg->H, cap = c->orb.max_out;
    if ((rc = ensure(c, c->img_f32, sizeof(float) * npix * B))) return rc;
    if ((rc = ensure(c, c->img_u8, npix * B))) return rc;
    if ((rc = ensure(c, c->minmax, 8 * B + 64))) return rc;
    // working copies owned by the batched path alone (host entry points on the same context never touch them, so the slice
    // carried from batch to batch survives an interleaved eorb_orb_extract / matcher call): slot 0 = previous batch's last slice
    if ((rc = ensure(c, c->fe_prev_kp, sizeof(eorb_keypoint) * cap * (B + 1)))) return rc;
    if ((rc = ensure(c, c->fe_prev_desc, 32 * cap * (B + 1)))) return rc;
    if ((rc = ensure(c, c->fe_prev_n, sizeof(int32_t) * (2 * B + 4)))) return rc;
    if ((rc = ensure(c, c->m_h, sizeof(int32_t) * cap * B))) return rc;   // matches12 (when the caller passes none)
    if ((rc = ensure(c, c->m_j, sizeof(int32_t) * (B + 1)))) return rc;    // nmatches
    EORB_HIP(c, hipMemsetAsync(c->fe_prev_n.p, 0, sizeof(int32_t) * (2 * B + 4), c->stream));
    c->fe_configured = true;
    c->fe_has_prev = false;
    return EORB_OK;
}

// the batch's user-visible records and the slice carried to the next batch, in one launch (six device-to-device copies cost a
// 3.6 ms step 37 us of launch gaps): segment s copies n[s] bytes (multiples of 4) from src[s] to dst[s]
struct CopySegs { void* dst[6]; const void* src[6]; size_t n[6]; int count; };
__global__ __launch_bounds__(256) void fe_publish_kernel(CopySegs S)
{
    for (int s = 0; s < S.count; s++) {
        const size_t n = S.n[s];
        char* d = (char*)S.dst[s]; const char* q = (const char*)S.src[s];
        if (!(((uintptr_t)d | (uintptr_t)q | n) & 15)) {
            const size_t n16 = n >> 4;
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) ((uint4*)d)[i] = ((const uint4*)q)[i];
        } else {
            const size_t n4 = n >> 2;
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) ((uint32_t*)d)[i] = ((const uint32_t*)q)[i];
        }
    }
}

static int fe_run_batch_common(eorb_ctx* c, const void* d_events, int raw, const int64_t* h_offsets, int B,
                               uint8_t* d_images, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                               int32_t* d_matches12, int32_t* d_nmatches)
{
    if (!c) return EORB_E_ARG;
    if (!c->fe_configured) return set_err(c, EORB_E_NOTCONF, "fe_run_batch: eorb_fe_configure not called");
    const eorb_fe_config& f = c->fe;
    const bool from_images = raw < 0;                    // (eorb_fe_run_batch_images_dev: the frames are given, nothing to accumulate)
    if (B < 1 || B > f.max_batch || (!h_offsets && !from_images)) return set_err(c, EORB_E_ARG, "fe_run_batch: bad batch size %d", B);
    if (from_images && !d_images) return set_err(c, EORB_E_ARG, "fe_run_batch_images: no images");
    fe_enter(c);
    const size_t npix = (size_t)f.W * f.H, cap = c->orb.max_out;
    uint8_t* img = d_images ? d_images : (uint8_t*)c->img_u8.p;
    // working copies: slot 0 of kp/desc/n holds the last slice of the previous batch (frame-to-frame matching)
    eorb_keypoint* wk = (eorb_keypoint*)c->fe_prev_kp.p;
    uint8_t* wd = (uint8_t*)c->fe_prev_desc.p;
    int32_t* wn = (int32_t*)c->fe_prev_n.p;             // [0] prev, [1..B] this batch, then mono index
    // (the images' normalisation to u8 is left to the extraction's first kernel, which builds level 0 from the float images)
    int rc = from_images ? EORB_OK : ev_accumulate_dev(c, d_events, raw, h_offsets, B, f.W, f.H, f.sigma, f.pol, 0, (float*)c->img_f32.p, img, 0,
                                                       (uint32_t*)c->minmax.p);
    if (rc) return rc;
    if (!from_images) { c->pyr0_f32 = (const float*)c->img_f32.p; c->pyr0_mm = (const uint32_t*)c->minmax.p; }
    rc = orb_extract_dev(c, img, f.W, npix, B, f.lap0, f.lap1, f.want_desc, wk + cap, wd + 32 * cap, nullptr, wn + 1,
                         wn + 1 + f.max_batch + 1);
    if (rc) return rc;
    if (f.match && f.want_desc) {
        eorb_grid_bounds gb;
        gb.minX = 0.f; gb.minY = 0.f; gb.maxX = (float)f.W; gb.maxY = (float)f.H;           // Frame.cc:862-866
        gb.invW = (float)kGridCols / (gb.maxX - gb.minX); gb.invH = (float)kGridRows / (gb.maxY - gb.minY);
        const int first = c->fe_has_prev ? 0 : 1;                  // pair p: slice p-1 (slot p) vs slice p (slot p+1)
        const int npairs = B - first;
        int32_t* m12 = d_matches12 ? d_matches12 : (int32_t*)c->m_h.p;
        int32_t* nm = d_nmatches ? d_nmatches : (int32_t*)c->m_j.p;
        if (!c->fe_has_prev) {
            EORB_HIP(c, hipMemsetAsync(nm, 0, sizeof(int32_t), c->stream));
            EORB_HIP(c, hipMemsetAsync(m12, 0xff, sizeof(int32_t) * cap, c->stream));
        }
        rc = search_init_dev(c, npairs, wk + cap * first, wn + first, cap, wd + 32 * cap * first, 32, 32 * cap, nullptr,
                             wk + cap * (first + 1), wn + first + 1, cap, wd + 32 * cap * (first + 1), 32, 32 * cap, nullptr,
                             (int)cap, (int)cap, gb, nullptr, m12 + cap * first, f.windowSize, f.nnratio, f.checkOri, nm + first);
        if (rc) return rc;
    }
    // user-visible outputs
    // (the carried slice -- the last one into slot 0 for the next batch -- never overlaps what it is copied from: slot B, B >= 1)
    CopySegs S; S.count = 0;
    auto seg = [&](void* d, const void* q, size_t n) { if (d && n) { S.dst[S.count] = d; S.src[S.count] = q; S.n[S.count] = n; S.count++; } };
    seg(d_kps, wk + cap, sizeof(eorb_keypoint) * cap * B);
    seg(d_desc, wd + 32 * cap, 32 * cap * B);
    seg(d_nkps, wn + 1, sizeof(int32_t) * B);
    seg(wk, wk + cap * B, sizeof(eorb_keypoint) * cap);
    seg(wd, wd + 32 * cap * B, 32 * cap);
    seg(wn, wn + B, sizeof(int32_t));
    fe_publish_kernel<<<512, 256, 0, c->stream>>>(S);
    EORB_LAUNCH_CHECK(c, "fe_publish_kernel");
    c->fe_has_prev = true;
    return EORB_OK;
}

int eorb_fe_run_batch_raw4_dev(eorb_ctx* c, const eorb_raw_event4* d_events, const int64_t* h_offsets, int B,
                               uint8_t* d_images, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                               int32_t* d_matches12, int32_t* d_nmatches)
{
    if (c && !c->lut_w) return set_err(c, EORB_E_NOTCONF, "fe_run_batch_raw4: eorb_set_undistort_maps not called");
    return fe_run_batch_common(c, d_events, 3, h_offsets, B, d_images, d_kps, d_desc, d_nkps, d_matches12, d_nmatches);
}

int eorb_fe_run_batch_raw2_dev(eorb_ctx* c, const eorb_raw_event2* d_events, const int64_t* h_offsets, int B,
                               uint8_t* d_images, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                               int32_t* d_matches12, int32_t* d_nmatches)
{
    if (c && !c->lut_w) return set_err(c, EORB_E_NOTCONF, "fe_run_batch_raw2: eorb_set_undistort_maps not called");
    if (c && c->fe_configured && c->fe.pol) return set_err(c, EORB_E_ARG, "fe_run_batch_raw2: the 2-byte record carries no polarity");
    return fe_run_batch_common(c, d_events, 4, h_offsets, B, d_images, d_kps, d_desc, d_nkps, d_matches12, d_nmatches);
}

int eorb_fe_run_batch_dev(eorb_ctx* c, const eorb_event16* d_events, const int64_t* h_offsets, int B,
                          uint8_t* d_images, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                          int32_t* d_matches12, int32_t* d_nmatches)
{
    return fe_run_batch_common(c, d_events, 0, h_offsets, B, d_images, d_kps, d_desc, d_nkps, d_matches12, d_nmatches);
}

int eorb_fe_run_batch_raw_dev(eorb_ctx* c, const eorb_raw_event* d_events, const int64_t* h_offsets, int B,
                              uint8_t* d_images, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                              int32_t* d_matches12, int32_t* d_nmatches)
{
    if (c && !c->lut_w) return set_err(c, EORB_E_NOTCONF, "fe_run_batch_raw: eorb_set_undistort_maps not called");
    return fe_run_batch_common(c, d_events, 1, h_offsets, B, d_images, d_kps, d_desc, d_nkps, d_matches12, d_nmatches);
}

int eorb_fe_last_f32_dev(eorb_ctx* c, const float** d_f32, float* h_minmax, int B)
{
    if (!c) return EORB_E_ARG;
    if (!c->fe_configured || !c->img_f32.p) return set_err(c, EORB_E_NOTCONF, "fe_last_f32: eorb_fe_configure not called");
    if (B < 0 || B > c->fe.max_batch) return set_err(c, EORB_E_ARG, "fe_last_f32: bad batch size %d", B);
    if (d_f32) *d_f32 = (const float*)c->img_f32.p;
    if (h_minmax && B) {
        fe_enter(c);
        std::vector<uint32_t> enc(2 * (size_t)B);
        { const int rc_dn = down(c, enc.data(), c->minmax.p, sizeof(uint32_t) * enc.size()); if (rc_dn) return rc_dn; }
        EORB_HIP(c, fe_stream_sync(c));
        for (size_t k = 0; k < enc.size(); k++) {           // the order-preserving integer encoding of the gather kernels' atomics
            const uint32_t e = enc[k], u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
            memcpy(&h_minmax[k], &u, 4);
        }
    }
    return EORB_OK;
}

int eorb_fe_run_batch_images_dev(eorb_ctx* c, const uint8_t* d_images, int B, eorb_keypoint* d_kps, uint8_t* d_desc, int32_t* d_nkps,
                                 int32_t* d_matches12, int32_t* d_nmatches)
{
    return fe_run_batch_common(c, nullptr, -1, nullptr, B, const_cast<uint8_t*>(d_images), d_kps, d_desc, d_nkps, d_matches12, d_nmatches);
}

}  // extern "C"
