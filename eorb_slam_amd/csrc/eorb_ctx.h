// eorb_ctx.h -- internal context shared by the translation units of libeorb_fe.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/eorb_fe.h"

namespace eorb {

constexpr int kMaxLevels = 16;
constexpr int kTile = 8;            // accumulation tile edge (one wavefront = 8x8 pixels)
constexpr int kChunk = 4096;        // events per binning chunk (one wavefront bins one chunk)
constexpr int kGridCols = 64;       // FRAME_GRID_COLS include/Frame.h:46
constexpr int kGridRows = 48;       // FRAME_GRID_ROWS include/Frame.h:45

struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
};

struct ProfEntry {
    std::string name;
    double total_ms = 0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct ChunkDesc {       // one binning chunk = a run of <= kChunk consecutive events of one slice
    int64_t start;       // index of the first event (into the batch's event array)
    int32_t n;
    int32_t slice;
};

// geometry of one pyramid level (ComputePyramid / ComputeKeyPointsOctTree, ORBextractor.cc:784-808,1240-1265)
struct LevelGeom {
    int w, h;            // level (ROI) size
    int bw, bh;          // bordered buffer size (w+2E, h+2E)
    int buf_off;         // byte offset of the bordered buffer inside a slice's pyramid block
    int roi_off;         // byte offset of the un-bordered (w*h) planes (score / blur) inside a slice block
    int minBX, minBY, maxBX, maxBY;
    int nCols, nRows, wCell, hCell;
    int cell_off;        // index of this level's first cell in the per-slice cell arrays
    int cand_cap;        // candidate capacity of the level
    int cand_off;        // offset (entries) of the level's candidate array inside a slice block
    int nfeat;           // mnFeaturesPerLevel
    int kp_cap;          // nfeat + slack
    int kp_off;          // offset (entries) of the level's keypoint array inside a slice block
    int xtab_off, ytab_off;   // resize coefficient tables (int offsets into the table buffer)
    int xmax;            // first dx whose source column is clamped (HResize)
    float scale;         // mvScaleFactor[level]
    int patch_size;      // (int)(31*scale)
    int node_cap;        // octree node pool capacity
};

struct OrbState {
    bool configured = false;
    eorb_orb_params p{};
    int W = 0, H = 0, edge = 0, nlevels = 0;
    float sf[kMaxLevels]{}, inv_sf[kMaxLevels]{};
    int nfeat[kMaxLevels]{};
    int umax[16]{};
    LevelGeom lv[kMaxLevels]{};
    int pyr_bytes = 0;       // per slice: all bordered level buffers
    int roi_bytes = 0;       // per slice: all un-bordered level planes
    int ncells = 0;          // per slice: cells over all levels
    int cell_cap = 0;        // candidates per cell (capacity)
    int cand_total = 0;      // per slice: sum of cand_cap
    int kp_total = 0;        // per slice: sum of kp_cap  (= eorb_orb_max_keypoints)
    int max_out = 0;
    int oct_lds[2] = {0, 0};       // dynamic LDS bytes of the octree kernel, per placement (single frames / many workgroups)
    int oct_direct_cap[2] = {0, 0};
    int oct_dyn[2] = {0, 0}, oct_dyn_lds[2] = {0, 0};   // dynamic placement in use (its direct-pass cap, -1: none), its LDS bytes
    int oct_all_lds[2] = {0, 0};   // every item of that placement is in LDS: the kernel variant with LDS-typed pointers
    int oct_scratch[2] = {0, 0};   // per (slice, level) global scratch bytes
    DevBuf tabs;             // resize tables (short/int), level geometry, pattern, umax
    DevBuf geom;
};

}  // namespace eorb

struct eorb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    bool prof = false;
    std::string prof_only;                       // ",name,name,": only these scopes are timed (empty: all)
    std::vector<eorb::ProfEntry> profs;
    std::vector<hipEvent_t> ev_pool;

    // accumulation workspaces
    eorb::DevBuf ev16, chunks, segoff, entries, img_f32, img_u8, minmax, tile_order, order_hist;
    // raw sensor events: undistortion maps (float2 per sensor pixel) and the tables derived from them
    eorb::DevBuf lut, src_info, stamps;
    eorb::DevBuf ev_info, ev_stamps;              // float events of the per-slice calls: the same tables per EVENT (ev_direct_slices_dev)
    // slot form of the raw accumulation (ev_slots.hip): per sensor pixel its tiles / slot numbers, per tile its rows; valid when sl_ok
    eorb::DevBuf sl_tab, sl_tile, sl_rows, sl_trace; long long sl_trace_n = 0;
    // per-batch workspaces of the slot form, one set per part of a batch (a batch of 64 slices or more runs as two halves)
    struct SlotWS { eorb::DevBuf chunks, segoff, entries, tile_order, plan, hot, rec16; };
    SlotWS sl_ws[2];
    // its streams (made together, ev_slots.hip sl_streams): sl_side = the register-row kernel (high priority), sl_pstream = the plan,
    // sl_gstream = the LDS gather of one half while the next half is binned; five events per part
    static constexpr int kSlotEvents = 10;
    hipStream_t sl_side = nullptr, sl_pstream = nullptr, sl_gstream = nullptr; hipEvent_t sl_ev[kSlotEvents] = {};
    int sl_last_parts = 0, sl_last_nb[2] = {0, 0};      // the last slot-form call: its parts, (slices x tiles) of each
    int sl_ok = 0, sl_null = 0, sl_rank_ok = -1;
    int ncu = 0;                                 // compute units of c->device (per context: a second context may sit on another GPU)
    unsigned sl_attr = 0;                        // bit per kernel instantiation whose dynamic-LDS opt-in was made on c->device
    int sl_last_rank = -1, sl_last_chunk = 0;    // the scatter the last slot-form call ran (1 rank form, 0 ballot form), its chunk size
    // pinned staging of the entry points that move their buffers one by one (up() / down() of the KeyFrame-side matchers, BoW, LK): bump
    // allocators, emptied by the entry's own stream wait; dn_pending: host destinations filled from dn_pin at that wait
    struct PinBump { void* p = nullptr; size_t cap = 0, used = 0; };
    PinBump up_pin, dn_pin;
    struct PendingDown { void* dst; size_t off, bytes; };
    std::vector<PendingDown> dn_pending;
    struct CopySeg { void* dst; const void* src; size_t n; };
    std::vector<CopySeg> up_queue, dn_queue;           // staged copies not launched yet: one multi-segment kernel per up_flush() / stream wait
    int* rb_pinned = nullptr;                   // 64 ints of pinned host memory: the landing place of small read-backs (position count, slot info)
    int sl_launched = 0; int sl_hinfo[6] = {0, 0, 0, 0, 0, 0};      // assignment kernels launched, their read-back (ev_slots_prepare_launch / _finish)
    int dd_src_info_done = 0;                   // float bulk path: src_info of the per-call table already launched
    long long sl_calls = 0; size_t sl_info_off = 0;     // test hook counters (eorb_debug_counter)
    // float events in bulk: the distinct positions of a call become the rows of a per-call stamp table (ev_accumulate_dev)
    eorb::DevBuf dd_tab, dd_src_info, dd_stamps, dd_ev, dd_cnt, dd_sl_tab, dd_sl_tile, dd_sl_rows;
    int dd_sl_ok = 0, dd_sl_null = 0; size_t dd_sl_info_off = 0;
    // ... and across calls: once a call's positions are known, the next calls of the context look theirs up in a frozen dictionary
    // (compact hash -> dense id < 65 536, tables per id) and only fall back to the per-call tabulation when a new position turns up
    eorb::DevBuf pd_hash, pd_lut, pd_src_info, pd_sl_tab, pd_sl_tile, pd_sl_rows, pd_cnt;
    int pd_valid = 0, pd_K = 0, pd_W = 0, pd_H = 0, pd_sl_null = 0, pd_cooldown = 0; float pd_sigma = 0.f; size_t pd_sl_info_off = 0;
    int dd_keep = 0;                             // the per-call position table holds an earlier call's positions and is extended, not cleared
    long long pd_hits = 0, pd_misses = 0;        // calls served by the dictionary / calls that found a new position (eorb_debug_counter)
    eorb::DevBuf focus_sd;                       // measureImageFocus: per-patch deviations
    int64_t dbg_dd_min = (int64_t)1 << 20;       // events from which the positions are deduplicated (test hook: "dedupe_min_events")
    int lut_w = 0, lut_h = 0, lut_check = 1;
    int lut_key_W = -1, lut_key_H = -1, lut_key_mode = -1; float lut_key_sigma = -1.f;
    // extractor workspaces
    eorb::OrbState orb;
    eorb::DevBuf pyr, score, blur, cell_cnt, cell_cand, lvl_cnt, lvl_kp, kp_angle, out_kp, out_desc, out_oob,
        out_n, oct_scratch, in_img;
    // matcher workspaces
    eorb::DevBuf m_a, m_b, m_c, m_d, m_e, m_f, m_g, m_h, m_i, m_j;
    eorb::DevBuf win_ws;                 // candidate lists of the two-phase window matchers
    eorb::DevBuf win_total;              // their per-pair entry counters: zero between calls (phase 2 puts its pair's back), win_total_n of them known to be
    size_t win_total_n = 0;
    unsigned win_attr_done = 0;          // bit KIND: the window matchers' kernels of that kind have their LDS opt-in on this context's device
    eorb::DevBuf arena;                  // host-buffer entry points: all inputs / outputs of one call, one H2D and one D2H copy
    void* dl_pinned = nullptr; size_t dl_cap = 0;      // pinned landing buffer of the D2H copy (the call synchronises before reading it)
    hipEvent_t dl_event = nullptr;                     // recorded behind that copy: what the call waits for
    bool mm_preset = false;                            // the next ev_accumulate_dev finds its running extremes initialised (per-slice calls)
    // pyramidal LK workspaces; klt_ref_key: the reference frame whose pyramid and derivatives the buffers hold (0 = none)
    eorb::DevBuf klt_pyr, klt_der, klt_scratch;
    unsigned long long klt_ref_key = 0, klt_ref_geo = 0, klt_ref_serial = 0;
    // L1 chain (eorb_ev_slice_extract / eorb_ev_slice_track): the tracker's reference image and points stay on the device
    eorb::DevBuf l1_ref_img, l1_ref_pts;
    int l1_nref = -1, l1_W = 0, l1_H = 0;
    size_t l1_img_off = 0; unsigned long long arena_gen = 0, l1_img_gen = 0;      // the last slice's u8 image inside the arena (eorb_ev_slice_image)
    // DBoW2 vocabulary (device copy) for eorb_bow_transform
    eorb::DevBuf voc;
    int voc_nnodes = 0, voc_L = 0; size_t voc_off[5] = {0, 0, 0, 0, 0};
    // batched front end
    bool fe_configured = false;
    eorb_fe_config fe{};
    eorb::DevBuf fe_prev_kp, fe_prev_desc, fe_prev_n, fe_pm;
    bool fe_has_prev = false;
    // pinned staging: a ring of slots, each guarded by an event recorded right after the H2D copy that reads it, so that
    // back-to-back un-synchronised *_dev calls never overwrite a slot whose copy is still queued
    static constexpr int kPinnedSlots = 4;
    struct PinnedSlot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false, lazy = false; };
    PinnedSlot pinned[kPinnedSlots];
    int pinned_next = 0, pinned_cur = -1;
    // sticky device status word (bits OR-ed by kernels of the *_dev paths: candidate / node-pool / keypoint overflow);
    // read back by eorb_sync / eorb_fe_status
    eorb::DevBuf status;
    // test hooks (eorb_debug_option): shrink the octree node pool to force an overflow; force the octree's global-memory layout
    // set by a caller whose float event images (and running extremes) are still to be normalised: orb_extract_dev's first kernel does
    // it while it builds level 0 (and writes the u8 images to its d_img argument); cleared by that call
    const float* pyr0_f32 = nullptr; const uint32_t* pyr0_mm = nullptr;
    int dbg_pool_shrink = 0, dbg_force_global = 0, dbg_oct_list = 0;      // (dbg_oct_list: the octree's list algorithm for every pass; applies at the next eorb_orb_configure)
    int dbg_gather_form = 0;                     // raw Gaussian accumulation: 0 by batch shape, 1 K2r, 2 K2s, 3 K2d (<= 4 slices), 4 slot lists (K2p)
    int dbg_orb_three_launches = 0;              // extraction: orientation, descriptors and output order as three kernels (the path of a lapping area) for every call
    int dbg_win_lds_ents = 0;                    // window matchers: entries phase 2 stages in LDS (to force its reads from global memory)
    int dbg_win_wcap = 0, dbg_win_ecap = 0;      // window matchers: list capacity per query / pool per pair (to force the full-scan path)
    int dbg_slot_rank = -1;                      // slot form: -1 by the device check / EORB_SLOT_RANK, 0 ballot scatter, 1 rank scatter (if the check passed)
    long long dbg_slot_hot_min = -1;             // slot form: list length from which the register-row kernel takes a list (-1: default / EORB_SLOT_HOT_MIN)
    int dbg_slot_hot_cap = 0;                    // slot form: lists per length bucket of the register-row kernel (0: kHotCap), to force the overflow branch
    int dbg_slot_hot_waves = 0;                  // slot form: wavefronts of the register-row kernel (0: default / EORB_SLOT_HOT_WAVES)
    int dbg_slot_prerank = -1;                   // slot form: ranks kept by the count pass, sl_scatter_pre_kernel (-1: default / EORB_SLOT_PRERANK, 0 / 1: test hook)
    int dbg_pd = 1;                              // float events in bulk: 0 = never freeze / use the position dictionary (test hook "position_dict")
    int dbg_slot_halves = -1;                    // slot form: -1 by the batch's shape / EORB_SLOT_HALVES, 0 one part, 1 two halves whatever the shape
};

namespace eorb {

int  set_err(eorb_ctx* c, int code, const char* fmt, ...);
int  ensure(eorb_ctx* c, DevBuf& b, size_t bytes);
void* pinned(eorb_ctx* c, size_t bytes);        // next free slot of the pinned ring (waits for the slot's previous copy)
void pinned_commit(eorb_ctx* c, bool lazy = false);   // call right after the hipMemcpyAsync that reads the slot; lazy: no event (the caller waits for the stream anyway: pinned_release_lazy)
void pinned_release_lazy(eorb_ctx* c);          // everything enqueued before the caller's last wait has completed
int  hip_check(eorb_ctx* c, hipError_t e, const char* what);
int* readback_buf(eorb_ctx* c);                 // c->rb_pinned, allocated on first use (nullptr: allocation failed)

// scoped per-kernel timing (HIP events on the ctx stream) when profiling is enabled
struct ProfScope {
    eorb_ctx* c; int idx; hipEvent_t a = nullptr, b = nullptr; hipStream_t st = nullptr;
    ProfScope(eorb_ctx* c, const char* name, hipStream_t stream = nullptr);      // stream: where the scope's kernels run (default: the context's)
    ~ProfScope();
};

#define EORB_HIP(c, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return eorb::hip_check((c), e__, #call); } while (0)
#define EORB_LAUNCH_CHECK(c, what) do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return eorb::hip_check((c), e__, what); } while (0)

// ev_accum.hip
int ev_accumulate_dev(eorb_ctx* c, const void* d_events, int raw, const int64_t* h_offsets, int B, int W, int H,
                      float sigma, int pol, int mode_count, float* d_f32, uint8_t* d_u8, int normalized,
                      uint32_t* d_minmax_enc);
int ev_direct_slices_dev(eorb_ctx* c, const void* d_events, int raw, const int64_t* beg, const int64_t* end, int B, int W, int H,
                         float sigma, int pol, float* d_f32, uint8_t* d_u8, int normalized, uint32_t* d_minmax_enc, bool mm_preset = false);
int ev_undistort_dev(eorb_ctx* c, const eorb_raw_event* d_raw, size_t n, int W, int H, double tsFactor, eorb_event* d_out, uint32_t* d_blk);
int ev_parse_text_dev(eorb_ctx* c, const char* d_text, size_t nbytes, uint64_t* d_lineend, eorb_raw_event* d_ev, uint8_t* d_status,
                      eorb_raw_event* d_out, uint32_t* d_blk, size_t max_lines, uint32_t h_res[3]);
// ev_slots.hip
int ev_slots_prepare_launch(eorb_ctx* c, int W, int H, int h, int TX, int TY);
int ev_slots_prepare(eorb_ctx* c, int W, int H, int h, int TX, int TY, const float* d_stamps /* nullptr: taps from c->lut */, int stamp_stride, int SWP,
                      float two_sig2, float norm);
struct SlotDict {              // float events looked up in the context's position dictionary by the count pass (ev_slots.hip)
    const uint4* hash; uint32_t mask; uint32_t* rec; int* miss;
    int rec2;                  // the dictionary holds fewer than 65 535 positions: rec is written as 2-byte records (0xffff = none), else 4-byte
};
int ev_slots_accumulate(eorb_ctx* c, const void* d_events, int stride, const int64_t* h_offsets, int B, int W, int H, int TX, int TY,
                        float* d_f32, uint32_t* d_minmax_enc, const SlotDict* dict = nullptr);
int ev_slots_trace_read(eorb_ctx* c, unsigned long long* out, long long max_records);
// klt.hip
int klt_track_dev(eorb_ctx* c, const uint8_t* d_prev, const uint8_t* d_next, int W, int H, int stride, const float* d_prev_pts,
                  float* d_next_pts, int n, int win, int maxLevel, int maxCount, double epsilon, int flags, float minEig,
                  uint8_t* d_status, float* d_err, unsigned long long ref_key = 0);
// ref_key != 0: d_prev is the reference frame `ref_key` -- its pyramid and derivatives are built once and reused while the key stays
// orb_extract.hip
int stereo_match_dev(eorb_ctx* c, const eorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, float mb, float mbf,
                     float* d_uright, float* d_depth, int32_t* d_sad, int32_t* d_nmatch);
int orb_extract_dev(eorb_ctx* c, const uint8_t* d_img, int img_stride, size_t img_slice_bytes, int B, int lap0, int lap1,
                    int want_desc, eorb_keypoint* d_kps, uint8_t* d_desc, uint8_t* d_oob, int32_t* d_n, int32_t* d_mono,
                    int32_t* d_flag_out = nullptr);      // d_flag_out: receives the overflow flag of this extraction (host entry point)
// match.hip
int search_init_dev(eorb_ctx* c, int npairs,
                    const eorb_keypoint* kps1, const int32_t* n1, size_t kp1_stride, const uint8_t* desc1, int dstride1, size_t desc1_slice,
                    const uint8_t* is_orb1,
                    const eorb_keypoint* kps2, const int32_t* n2, size_t kp2_stride, const uint8_t* desc2, int dstride2, size_t desc2_slice,
                    const uint8_t* is_orb2, int cap1, int cap2,
                    eorb_grid_bounds gb, float* prev_matched, int32_t* matches12, int windowSize, float nnratio,
                    int checkOri, int32_t* nmatches);

}  // namespace eorb
