// eorb_host.hpp -- C++ host side above the C ABI (include/eorb_fe.h), mirroring the reference's three seams
// with the same names, argument meaning and error behaviour, but OpenCV-free (own minimal Mat / KeyPoint):
//
//   EORB_SLAM::EvImConverter::ev2im / ev2im_gauss     include/Event/EventConversion.h:52-56
//   ORB_SLAM3::ORBextractor::operator()                include/ORBextractor.h:75-81 (both overloads)
//   ORB_SLAM3::ORBmatcher::SearchForInitialization      include/ORBmatcher.h (ORBmatcher.cc:714-831)
//
// The reference's own build keeps cv::Mat / cv::KeyPoint: INTEGRATION.md shows that adapter.  This header is
// what a C++ host without OpenCV uses, and what tests/test_host_cpp.py compiles.  Everything runs on the GPU
// through libeorb_fe.so; there is no CPU path here.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/eorb_fe.h"

namespace eorb_host {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// one context per calling thread AT A TIME (the ABI's threading contract)
class Context {
public:
    explicit Context(int device = 0, void* hipStream = nullptr) {
        int rc = eorb_create(device, hipStream, &h_);
        if (rc != EORB_OK) throw Error(rc, "eorb_create failed");
    }
    ~Context() { eorb_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    eorb_ctx* get() const { return h_; }
    void check(int rc) const { if (rc != EORB_OK) throw Error(rc, eorb_last_error(h_)); }
    unsigned maps_epoch = 0, voc_epoch = 0;          // which shared state of the pool this context has loaded
private:
    eorb_ctx* h_ = nullptr;
};

// Warm contexts shared by the threads of the process.  The reference calls the converters from long-lived tracker threads AND
// from four transient std::threads per motion-compensated image (src/Event/EvImBuilder.cpp:1165-1193): a context per thread
// would be created (device workspaces, pinned staging) and destroyed on every dispatch.  A thread borrows a context on its first
// call and hands it back when it exits; state that the reference keeps in process-wide objects (the calibrator's undistortion maps,
// the vocabulary) is recorded in the pool and loaded into a borrowed context that has not seen its current version yet.
class ContextPool {
public:
    static ContextPool& instance() { static ContextPool p; return p; }
    Context* acquire() {
        Context* c = nullptr;
        {
            std::lock_guard<std::mutex> g(m_);
            if (!free_.empty()) { c = free_.back(); free_.pop_back(); }
        }
        if (!c) {
            std::unique_ptr<Context> n(new Context());
            c = n.get();
            std::lock_guard<std::mutex> g(m_);
            all_.push_back(std::move(n));
        }
        try { sync_state(*c); }
        catch (...) { release(c); throw; }           // (a failed upload must not strand the context outside the free list)
        return c;
    }
    void release(Context* c) { std::lock_guard<std::mutex> g(m_); free_.push_back(c); }
    size_t created() const { std::lock_guard<std::mutex> g(m_); return all_.size(); }
    // process-wide state: immutable snapshots, replaced as a whole
    struct Maps { std::vector<float> mapX, mapY; int LW = 0, LH = 0; bool check = true; };
    struct Voc { int L = 0; std::vector<int32_t> childOff, childIds, wordId; std::vector<uint8_t> nodeDesc; std::vector<double> weight; };
    void set_maps(const std::vector<float>& mapX, const std::vector<float>& mapY, int LW, int LH, bool check) {
        auto m = std::make_shared<Maps>(); m->mapX = mapX; m->mapY = mapY; m->LW = LW; m->LH = LH; m->check = check;
        std::lock_guard<std::mutex> g(m_);
        maps_ = std::move(m); maps_epoch_.store(maps_epoch_.load(std::memory_order_relaxed) + 1, std::memory_order_release);
    }
    void set_vocabulary(Voc v) {
        auto p = std::make_shared<Voc>(std::move(v));
        std::lock_guard<std::mutex> g(m_);
        voc_ = std::move(p); voc_epoch_.store(voc_epoch_.load(std::memory_order_relaxed) + 1, std::memory_order_release);
    }
    // Loads the maps / the vocabulary into a context that has not seen their current version.  Called on every API call of a
    // thread: the common case is two atomic loads; the process-wide mutex is only held to take a reference to the snapshot, never
    // across the GPU upload (other threads' calls do not wait for it).  `c` belongs to the calling thread.
    void sync_state(Context& c) {
        if (c.maps_epoch != maps_epoch_.load(std::memory_order_acquire)) {
            std::shared_ptr<const Maps> m; unsigned e;
            { std::lock_guard<std::mutex> g(m_); m = maps_; e = maps_epoch_.load(std::memory_order_relaxed); }
            if (m) c.check(eorb_set_undistort_maps(c.get(), m->mapX.data(), m->mapY.data(), m->LW, m->LH, m->check));
            c.maps_epoch = e;
        }
        if (c.voc_epoch != voc_epoch_.load(std::memory_order_acquire)) {
            std::shared_ptr<const Voc> v; unsigned e;
            { std::lock_guard<std::mutex> g(m_); v = voc_; e = voc_epoch_.load(std::memory_order_relaxed); }
            if (v) c.check(eorb_bow_set_vocabulary(c.get(), (int)v->childOff.size() - 1, v->L, v->childOff.data(), v->childIds.data(),
                                                   v->nodeDesc.data(), v->wordId.data(), v->weight.data()));
            c.voc_epoch = e;
        }
    }
private:
    mutable std::mutex m_;
    std::vector<std::unique_ptr<Context>> all_;
    std::vector<Context*> free_;
    std::shared_ptr<const Maps> maps_; std::atomic<unsigned> maps_epoch_{0};
    std::shared_ptr<const Voc> voc_; std::atomic<unsigned> voc_epoch_{0};
};

// the calling thread's context: borrowed from the pool for the lifetime of the thread
inline Context& thread_context() {
    struct Lease {
        Context* c;
        Lease() : c(ContextPool::instance().acquire()) {}
        ~Lease() { ContextPool::instance().release(c); }
    };
    thread_local Lease lease;
    ContextPool::instance().sync_state(*lease.c);        // (a long-lived thread picks up maps / vocabulary set after it started)
    return *lease.c;
}

// minimal stand-ins for cv::Mat (CV_8UC1 / CV_32FC1) and cv::KeyPoint
template <typename T> struct Mat_ {
    int rows = 0, cols = 0;
    std::vector<T> data;
    Mat_() = default;
    Mat_(int r, int c) : rows(r), cols(c), data((size_t)r * c) {}
    bool empty() const { return rows == 0 || cols == 0; }
    T* ptr(int r = 0) { return data.data() + (size_t)r * cols; }
    const T* ptr(int r = 0) const { return data.data() + (size_t)r * cols; }
};
using Mat8 = Mat_<uint8_t>;
using Mat32f = Mat_<float>;
using KeyPoint = eorb_keypoint;         // same 28-byte layout as cv::KeyPoint
using EventData = eorb_event;           // same 24-byte layout as EORB_SLAM::EventData

}  // namespace eorb_host

namespace EORB_SLAM {

// include/Event/EventConversion.h:45-75: static converters; thread-safe through per-thread contexts
struct EvImConverter {
    // returns the CV_8UC1 image when `normalized` (and, for ev2im, max > min), else the CV_32FC1 image in out32
    static bool ev2im(const std::vector<eorb_host::EventData>& vEvData, unsigned imWidth, unsigned imHeight, bool pol,
                      bool normalized, eorb_host::Mat8& out8, eorb_host::Mat32f& out32) {
        auto& c = eorb_host::thread_context();
        out8 = eorb_host::Mat8((int)imHeight, (int)imWidth); out32 = eorb_host::Mat32f((int)imHeight, (int)imWidth);
        int is_u8 = 0;
        c.check(eorb_ev2im(c.get(), vEvData.data(), vEvData.size(), (int)imWidth, (int)imHeight, pol, normalized,
                           out32.ptr(), out8.ptr(), nullptr, &is_u8));
        return is_u8 != 0;
    }
    static bool ev2im_gauss(const std::vector<eorb_host::EventData>& vEvData, unsigned imWidth, unsigned imHeight,
                            float sigma, bool pol, bool normalized, eorb_host::Mat8& out8, eorb_host::Mat32f& out32) {
        auto& c = eorb_host::thread_context();
        out8 = eorb_host::Mat8((int)imHeight, (int)imWidth); out32 = eorb_host::Mat32f((int)imHeight, (int)imWidth);
        c.check(eorb_ev2im_gauss(c.get(), vEvData.data(), vEvData.size(), (int)imWidth, (int)imHeight, sigma, pol,
                                 normalized, out32.ptr(), out8.ptr(), nullptr));
        return normalized;
    }
    // raw sensor events + MyCalibrator maps (EventLoader.cpp:264-305 fused with ev2im_gauss)
    static void ev2im_gauss_raw(const std::vector<eorb_raw_event>& vRaw, unsigned imWidth, unsigned imHeight, float sigma, bool pol,
                                bool normalized, eorb_host::Mat8& out8, eorb_host::Mat32f& out32) {
        auto& c = eorb_host::thread_context();
        out8 = eorb_host::Mat8((int)imHeight, (int)imWidth); out32 = eorb_host::Mat32f((int)imHeight, (int)imWidth);
        c.check(eorb_ev2im_gauss_raw(c.get(), vRaw.data(), vRaw.size(), (int)imWidth, (int)imHeight, sigma, pol, normalized,
                                     out32.ptr(), out8.ptr(), nullptr));
    }
};

// src/Event/EventLoader.cpp: the parts of EventDataStore the GPU takes over
struct EventDataStore {
    // MyCalibrator::mUndistMapX / mUndistMapY (LH x LW floats each) for this thread's context
    static void setUndistortMaps(const std::vector<float>& mapX, const std::vector<float>& mapY, int LW, int LH, bool checkInImage) {
        eorb_host::ContextPool::instance().set_maps(mapX, mapY, LW, LH, checkInImage);      // every context of the process
    }
    // getline + parseLine + isComment over a text buffer (:80-92); throws eorb_host::Error for a line outside the grammar
    static std::vector<eorb_raw_event> parseText(const std::string& text) {
        auto& c = eorb_host::thread_context();
        size_t lines = 1; for (char ch : text) lines += ch == '\n';
        std::vector<eorb_raw_event> out(lines); size_t n = 0; int64_t bad = -1;
        c.check(eorb_parse_events_text(c.get(), text.data(), text.size(), out.data(), out.size(), &n, &bad));
        out.resize(n);
        return out;
    }
    // the rectification loop of getEventChunkRectified (:264-305)
    static std::vector<eorb_host::EventData> rectify(const std::vector<eorb_raw_event>& raw, int imWidth, int imHeight, double tsFactor) {
        auto& c = eorb_host::thread_context();
        std::vector<eorb_host::EventData> out(raw.size()); size_t n = 0;
        c.check(eorb_undistort_events(c.get(), raw.data(), raw.size(), imWidth, imHeight, tsFactor, out.data(), &n));
        out.resize(n);
        return out;
    }
};

// The data path of EORB_SLAM::EvImBuilder::Track (src/Event/EvImBuilder.cpp:1300-1515) over the one-call seams: per chunk of
// l1ChunkSize events the event image and its frame -- INIT: detect-only ORBextractor + ELK_Tracker::setRefImage (init :568-592);
// TRACKING: ELK_Tracker::trackAndMatchCurrImage (KLT_Tracker.cpp:215-234) + refineTrackedPts (:104-151) -- the window-size rule
// (resolveEvWinSize :209-232) and, on a dispatch, generateMCImage (:1146-1247) + isMcImageGood (:260-267).  The optimisers that hand
// generateMCImage its poses, step()'s two-view refinement, the IMU and the L2 queue stay with the caller, as in the reference.
// (Python twin with the same members: eorb_slam_amd/frontend.py::EvImBuilder; tests/test_gpu_chain.py checks that one against the
// oracle's chain chunk by chunk, tests/test_host_cpp.py checks this one against the separate seams.)
class EvImBuilder {
public:
    enum TrackState { IDLE, INIT, TRACKING };
    struct Params {            // Event.* of Examples/Event/EvETHZ.yaml:178-208
        int imWidth = 240, imHeight = 180; unsigned l1ChunkSize = 2000; int l1NumLoop = 3; bool l1FixedWinSz = false, continTracking = true;
        float maxPixelDisp = 3.f, l1WinOverlap = 0.5f, l1ImSigma = 1.f; double minEvGenRate = 1.0;
        int maxNumPts = 400, fastTh = 0, imMargin = 9; eorb_klt_params klt{23, 1, 10, 0.03, 1e-4f};
    };
    struct MciPoses { const eorb_se3_motion* dp = nullptr; const eorb_se3_motion* ba = nullptr; const float* se2 = nullptr; int nse2 = 3; const eorb_camera* cam = nullptr; };
    struct ChunkResult {
        TrackState state = IDLE; bool dispatched = false; int skipped = 0;
        std::vector<eorb_host::KeyPoint> kps;                          // INIT frame
        std::vector<float> pts; std::vector<uint8_t> status; std::vector<float> err; std::vector<int> matches12;   // TRACKING frame
        unsigned nMatches = 0; float medPxDisp = 0.f; unsigned chunkSize = 0;
        float focus[5] = {-1, -1, -1, -1, -1}; int winner = -1; eorb_host::Mat8 mcImage; std::vector<eorb_host::KeyPoint> l2Kps; bool mcGood = false;
        size_t window = 0, overlap = 0;                                // events of the dispatched window / handed back to the queue (its tail)
    };
    static const int DEF_TH_MIN_KPTS = 100, DEF_TH_MIN_MATCHES = 50;   // include/Event/EventData.h:24-26

    explicit EvImBuilder(const Params& p) : P(p), mL1EvWinSize(p.l1ChunkSize), mInitL1EvWinSize(p.l1ChunkSize) {
        eorb_orb_params q{p.maxNumPts, 1.0f, 1, p.fastTh, 0, p.imMargin, p.imWidth};       // "FAST = ORB with one level": EvBaseTracker.cpp:150-164
        l1_.check(eorb_orb_configure(l1_.get(), &q, p.imWidth, p.imHeight));
        q.nfeatures = 2 * p.maxNumPts;                                  // the L2 tracker's extractor: EvAsynchTracker.cpp:51
        l2_.check(eorb_orb_configure(l2_.get(), &q, p.imWidth, p.imHeight));
        cap1_ = eorb_orb_max_keypoints(l1_.get()); cap2_ = eorb_orb_max_keypoints(l2_.get());
        mnWinOverlap = (unsigned long)(p.l1WinOverlap * (double)(p.l1NumLoop * mL1EvWinSize));      // :40
        reset();
    }
    void resetAll() { mStat = IDLE; mL1EvWinSize = mInitL1EvWinSize; reset(); }                      // :70-93
    unsigned getL1ChunkSize() const { return mL1EvWinSize; }

    // one pass of the loop body for the chunk l1Evs; `poses` = what the resolve*() calls of generateMCImage deliver (absent = failed)
    ChunkResult Track(const std::vector<eorb_host::EventData>& l1Evs, const MciPoses* (*mciPoses)(const std::vector<eorb_host::EventData>&, void*) = nullptr, void* user = nullptr) {
        ChunkResult out;
        if (mStat == IDLE) mStat = INIT;
        if (mStat == INIT) reset();
        if (l1Evs.empty()) return out;
        const double evTspan = l1Evs.back().ts - l1Evs[0].ts;           // calcEventGenRate, src/Event/EventData.cpp:14-19
        const double evGenRate = (double)l1Evs.size() / (evTspan * P.imWidth * P.imHeight);
        out.state = mStat;
        const int rate = checkEvGenRate(evGenRate);
        if (rate != 0) {
            if (rate == -1) mStat = INIT; else mvSharedL2Evs.insert(mvSharedL2Evs.end(), l1Evs.begin(), l1Evs.end());
            out.skipped = rate;
            return out;
        }
        bool sendMCF = false;
        if (mStat == INIT) {
            out.kps.assign(cap1_, eorb_host::KeyPoint{});
            int n = 0, mono = 0;
            l1_.check(eorb_ev_slice_extract(l1_.get(), l1Evs.data(), nullptr, l1Evs.size(), P.l1ImSigma, 0, 1000, 0, out.kps.data(), nullptr, nullptr,
                                            cap1_, &n, &mono, nullptr));
            out.kps.resize(n);
            mRefKPoints = out.kps; mvMatchesCnt.assign(n, 1);
            mLastTrackedPts.resize(2 * (size_t)n);
            for (int i = 0; i < n; i++) { mLastTrackedPts[2 * i] = out.kps[i].x; mLastTrackedPts[2 * i + 1] = out.kps[i].y; }
            if (n > DEF_TH_MIN_KPTS || (P.continTracking && P.l1FixedWinSz)) { updateState(l1Evs); mStat = TRACKING; }
            else mL1EvWinSize = mInitL1EvWinSize;
        } else {
            const int nref = (int)mRefKPoints.size();
            out.status.assign(nref, 0); out.err.assign(nref, 0.f);
            l1_.check(eorb_ev_slice_track(l1_.get(), l1Evs.data(), nullptr, l1Evs.size(), P.l1ImSigma, &P.klt, mLastTrackedPts.data(), out.status.data(),
                                          out.err.data(), nref, nullptr));
            out.pts = mLastTrackedPts;
            out.matches12.assign(nref, -1);
            std::vector<float> vPxDisp;
            for (int i = 0; i < nref; i++) {                            // refineTrackedPts
                const float x = out.pts[2 * i], y = out.pts[2 * i + 1];
                if (out.status[i] == 1 && x >= 0 && x < (float)P.imWidth && y >= 0 && y < (float)P.imHeight) {
                    mvMatchesCnt[i]++; out.matches12[i] = i; out.nMatches++;
                    const float dx = x - mRefKPoints[i].x, dy = y - mRefKPoints[i].y;
                    vPxDisp.push_back(std::sqrt(dx * dx + dy * dy));
                }
            }
            std::sort(vPxDisp.begin(), vPxDisp.end());
            out.medPxDisp = vPxDisp.empty() ? 0.f : vPxDisp[vPxDisp.size() / 2];
            if (out.nMatches < (unsigned)DEF_TH_MIN_MATCHES && !P.continTracking && mCurrIdx < 3) { mL1EvWinSize = mInitL1EvWinSize; mStat = INIT; out.chunkSize = mL1EvWinSize; return out; }
            updateState(l1Evs);
            const bool dispatchMCI = resolveEvWinSize(out.medPxDisp);
            if (dispatchMCI || out.nMatches < (unsigned)DEF_TH_MIN_MATCHES) { mStat = INIT; sendMCF = true; }
        }
        out.chunkSize = mL1EvWinSize;
        if (sendMCF) {
            const MciPoses none; const MciPoses* mp = mciPoses ? mciPoses(mvSharedL2Evs, user) : &none;
            out.mcImage = eorb_host::Mat8(P.imHeight, P.imWidth);
            out.l2Kps.assign(cap2_, eorb_host::KeyPoint{});
            int n = 0;
            l1_.check(eorb_ev_mc_contest(l1_.get(), mvSharedL2Evs.data(), mvSharedL2Evs.size(), mp->cam, mp->dp, mp->ba, mp->se2, mp->nse2, P.imWidth, P.imHeight,
                                         P.l1ImSigma, out.focus, &out.winner, out.mcImage.ptr(), l2_.get(), 0, 1000, out.l2Kps.data(), cap2_, &n));
            out.l2Kps.resize(n);
            out.dispatched = true; out.window = mvSharedL2Evs.size();
            out.mcGood = n > DEF_TH_MIN_KPTS || P.continTracking;
            if (P.continTracking) out.overlap = P.l1FixedWinSz ? mnWinOverlap : (size_t)(mvSharedL2Evs.size() * P.l1WinOverlap);     // :1465-1469
        }
        return out;
    }
    const std::vector<eorb_host::EventData>& accumulatedEvents() const { return mvSharedL2Evs; }

private:
    void reset() { mCurrIdx = 0; mCntLowEvGenRate = 0; mvSharedL2Evs.clear(); mvMatchesCnt.clear(); }          // :95-139
    void updateState(const std::vector<eorb_host::EventData>& ev) { mCurrIdx++; mvSharedL2Evs.insert(mvSharedL2Evs.end(), ev.begin(), ev.end()); }      // :427-435
    int checkEvGenRate(double rate) {                                    // :284-328
        if (rate > P.minEvGenRate) { mCntLowEvGenRate = 0; return 0; }
        if (P.continTracking) return (!P.l1FixedWinSz && mStat == INIT) ? -1 : 0;
        if (mStat == INIT) return -1;
        return ++mCntLowEvGenRate > 3 ? -1 : 1;
    }
    bool resolveEvWinSize(float medPxDisp) {                             // :209-232, calcNewL1ChunkSize :197-201
        if (!P.l1FixedWinSz && medPxDisp > P.maxPixelDisp) { mL1EvWinSize = (unsigned)floorf(((float)(mCurrIdx + 1) / medPxDisp) * ((float)mL1EvWinSize)); return true; }
        return P.l1FixedWinSz && (int)mCurrIdx >= P.l1NumLoop;
    }
    Params P;
    eorb_host::Context l1_, l2_;
    int cap1_ = 0, cap2_ = 0;
    TrackState mStat = IDLE;
    unsigned mCurrIdx = 0, mL1EvWinSize, mInitL1EvWinSize; int mCntLowEvGenRate = 0; unsigned long mnWinOverlap = 0;
    std::vector<eorb_host::EventData> mvSharedL2Evs;
    std::vector<eorb_host::KeyPoint> mRefKPoints; std::vector<float> mLastTrackedPts; std::vector<int> mvMatchesCnt;
};

}  // namespace EORB_SLAM

namespace ORB_SLAM3 {

struct ORBxParams {            // include/ORBextractor.h:33-47
    int nfeatures = 0; float scaleFactor = 1; int nlevels = 1; int iniThFAST = 10; int minThFAST = 7; int edgeTh = 19;
    int imWidth = 0, imHeight = 0;
};

class ORBextractor {           // include/ORBextractor.h:49-139; one instance per thread, like the reference
public:
    explicit ORBextractor(const ORBxParams& p) : p_(p) {
        eorb_orb_params q{p.nfeatures, p.scaleFactor, p.nlevels, p.iniThFAST, p.minThFAST, p.edgeTh, p.imWidth};
        ctx_.check(eorb_orb_configure(ctx_.get(), &q, p.imWidth, p.imHeight));
        cap_ = eorb_orb_max_keypoints(ctx_.get());
        mvScaleFactor.resize(p.nlevels); mvInvScaleFactor.resize(p.nlevels); mnFeaturesPerLevel.resize(p.nlevels);
        ctx_.check(eorb_orb_get_tables(ctx_.get(), mvScaleFactor.data(), mvInvScaleFactor.data(), mnFeaturesPerLevel.data(), &edge_));
    }
    // with descriptors (src/ORBextractor.cc:1092-1176); returns monoIndex, -1 for an empty image
    int operator()(const eorb_host::Mat8& image, std::vector<eorb_host::KeyPoint>& keypoints, eorb_host::Mat8& descriptors,
                   const std::vector<int>& vLappingArea) {
        if (image.empty()) return -1;
        keypoints.assign(cap_, eorb_host::KeyPoint{});
        eorb_host::Mat8 d(cap_, 32);
        int n = 0, mono = 0;
        ctx_.check(eorb_orb_extract(ctx_.get(), image.ptr(), image.cols, image.rows, image.cols, vLappingArea[0], vLappingArea[1], 1,
                                    keypoints.data(), d.ptr(), nullptr, cap_, &n, &mono));
        keypoints.resize(n);
        descriptors = eorb_host::Mat8(n, 32);            // released (0 rows) when n == 0, like _descriptors.release()
        if (n) std::memcpy(descriptors.ptr(), d.ptr(), (size_t)n * 32);
        return mono;
    }
    // detect only (:1178-1238)
    int operator()(const eorb_host::Mat8& image, std::vector<eorb_host::KeyPoint>& keypoints, const std::vector<int>& vLappingArea) {
        if (image.empty()) return -1;
        keypoints.assign(cap_, eorb_host::KeyPoint{});
        int n = 0, mono = 0;
        ctx_.check(eorb_orb_extract(ctx_.get(), image.ptr(), image.cols, image.rows, image.cols, vLappingArea[0], vLappingArea[1], 0,
                                    keypoints.data(), nullptr, nullptr, cap_, &n, &mono));
        keypoints.resize(n);
        return mono;
    }
    // Frame::Frame(imLeft, imRight, ...) (src/Frame.cc:97-152): ExtractORB on both images (:122-125) + ComputeStereoMatches (:869-1048)
    // in one call; mb = baseline (mbf / fx).  Fills what the constructor fills: mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight,
    // mvuRight, mvDepth.  Returns the number of correlated matches before the median cut.
    int ExtractStereo(const eorb_host::Mat8& imLeft, const eorb_host::Mat8& imRight, float mb, float mbf,
                      std::vector<eorb_host::KeyPoint>& mvKeys, eorb_host::Mat8& mDescriptors,
                      std::vector<eorb_host::KeyPoint>& mvKeysRight, eorb_host::Mat8& mDescriptorsRight,
                      std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
        mvKeys.assign(cap_, eorb_host::KeyPoint{}); mvKeysRight.assign(cap_, eorb_host::KeyPoint{});
        eorb_host::Mat8 dl(cap_, 32), dr(cap_, 32);
        mvuRight.assign(cap_, -1.0f); mvDepth.assign(cap_, -1.0f);
        int nl = 0, nr = 0, nm = 0;
        ctx_.check(eorb_frame_stereo(ctx_.get(), imLeft.ptr(), imRight.ptr(), imLeft.cols, imLeft.rows, imLeft.cols, mb, mbf, mvKeys.data(), dl.ptr(), &nl,
                                     mvKeysRight.data(), dr.ptr(), &nr, cap_, mvuRight.data(), mvDepth.data(), &nm));
        mvKeys.resize(nl); mvKeysRight.resize(nr); mvuRight.resize(nl); mvDepth.resize(nl);
        mDescriptors = eorb_host::Mat8(nl, 32); mDescriptorsRight = eorb_host::Mat8(nr, 32);
        if (nl) std::memcpy(mDescriptors.ptr(), dl.ptr(), (size_t)nl * 32);
        if (nr) std::memcpy(mDescriptorsRight.ptr(), dr.ptr(), (size_t)nr * 32);
        return nm;
    }
    int GetLevels() const { return p_.nlevels; }
    float GetScaleFactor() const { return p_.scaleFactor; }
    std::vector<float> GetScaleFactors() const { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() const { return mvInvScaleFactor; }
    int GetNumFeatures() const { return p_.nfeatures; }
    int GetEdgeThreshold() const { return edge_; }
    std::vector<float> mvScaleFactor, mvInvScaleFactor;
    std::vector<int> mnFeaturesPerLevel;
private:
    ORBxParams p_;
    eorb_host::Context ctx_;
    int cap_ = 0, edge_ = 0;
};

// the part of a Frame the matchers read (undistorted keypoints, descriptors, image bounds)
struct FrameView {
    const std::vector<eorb_host::KeyPoint>* kps; const eorb_host::Mat8* desc; eorb_grid_bounds gb;
    FrameView(const std::vector<eorb_host::KeyPoint>& k, const eorb_host::Mat8& d, int W, int H) : kps(&k), desc(&d) {
        gb.minX = 0.f; gb.minY = 0.f; gb.maxX = (float)W; gb.maxY = (float)H;        // Frame.cc:862-866
        gb.invW = 64.f / (gb.maxX - gb.minX); gb.invH = 48.f / (gb.maxY - gb.minY);  // Frame.cc:362-363
    }
    int numAllKPts() const { return (int)kps->size(); }
};

class ORBmatcher {             // include/ORBmatcher.h:40-116
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
    explicit ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    // vbPrevMatched: (x,y) per F1 keypoint, updated in place; vnMatches12 resized to F1.numAllKPts()
    int SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<float>& vbPrevMatched,
                                std::vector<int>& vnMatches12, int windowSize = 10) {
        auto& c = eorb_host::thread_context();
        vnMatches12.assign(F1.numAllKPts(), -1);
        int nm = 0;
        c.check(eorb_search_for_initialization(c.get(), F1.kps->data(), F1.numAllKPts(), F1.desc->ptr(), F1.desc->cols, nullptr,
                                               F2.kps->data(), F2.numAllKPts(), F2.desc->ptr(), F2.desc->cols, nullptr, &F2.gb,
                                               vbPrevMatched.data(), vnMatches12.data(), windowSize, mfNNratio,
                                               mbCheckOrientation, &nm));
        return nm;
    }
    // DBoW2::FeatureVector as CSR (nodes ascending, node_off, idx) -- what ORBVocabulary::transform returns below
    struct FeatureVector { std::vector<uint32_t> nodes; std::vector<int32_t> off{0}, idx; };
    // SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (:276-478): match_f[j] = KeyFrame feature whose map point goes to F's j
    int SearchByBoW(const FrameView& KF, const std::vector<uint8_t>& kfHasMP, const FeatureVector& kfFV, const FrameView& F,
                    const FeatureVector& fFV, std::vector<int>& match_f) {
        auto& c = eorb_host::thread_context();
        match_f.assign(F.numAllKPts(), -1); int nm = 0;
        c.check(eorb_search_by_bow(c.get(), KF.kps->data(), KF.numAllKPts(), KF.desc->ptr(), kfHasMP.data(), kfFV.nodes.data(),
                                   kfFV.off.data(), kfFV.idx.data(), (int)kfFV.nodes.size(), F.kps->data(), F.numAllKPts(), F.desc->ptr(),
                                   fFV.nodes.data(), fFV.off.data(), fFV.idx.data(), (int)fFV.nodes.size(), match_f.data(),
                                   mfNNratio, mbCheckOrientation, &nm));
        return nm;
    }
    // SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (:833-973)
    int SearchByBoW(const FrameView& KF1, const std::vector<uint8_t>& hasMP1, const FeatureVector& FV1, const FrameView& KF2,
                    const std::vector<uint8_t>& hasMP2, const FeatureVector& FV2, std::vector<int>& match12, bool) {
        auto& c = eorb_host::thread_context();
        match12.assign(KF1.numAllKPts(), -1); int nm = 0;
        c.check(eorb_search_by_bow_kf(c.get(), KF1.kps->data(), KF1.numAllKPts(), KF1.desc->ptr(), hasMP1.data(), FV1.nodes.data(),
                                      FV1.off.data(), FV1.idx.data(), (int)FV1.nodes.size(), KF2.kps->data(), KF2.numAllKPts(),
                                      KF2.desc->ptr(), hasMP2.data(), FV2.nodes.data(), FV2.off.data(), FV2.idx.data(),
                                      (int)FV2.nodes.size(), match12.data(), mfNNratio, mbCheckOrientation, &nm));
        return nm;
    }
    // SearchForTriangulation (:975-1214, mono): ep / F12 / level tables from the caller (see include/eorb_fe.h)
    int SearchForTriangulation(const FrameView& KF1, const std::vector<uint8_t>& elig1, const FeatureVector& FV1, const FrameView& KF2,
                               const std::vector<uint8_t>& elig2, const FeatureVector& FV2, const float ep[2], const float F12[9],
                               const std::vector<float>& scale2, const std::vector<float>& sigma2_2,
                               std::vector<std::pair<size_t, size_t>>& vMatchedPairs, bool bCoarse = false) {
        auto& c = eorb_host::thread_context();
        std::vector<int> m12(KF1.numAllKPts(), -1); int nm = 0;
        c.check(eorb_search_for_triangulation(c.get(), KF1.kps->data(), KF1.numAllKPts(), KF1.desc->ptr(), KF1.desc->cols, elig1.data(),
                                              FV1.nodes.data(), FV1.off.data(), FV1.idx.data(), (int)FV1.nodes.size(), KF2.kps->data(),
                                              KF2.numAllKPts(), KF2.desc->ptr(), KF2.desc->cols, elig2.data(), FV2.nodes.data(),
                                              FV2.off.data(), FV2.idx.data(), (int)FV2.nodes.size(), ep, F12, scale2.data(),
                                              sigma2_2.data(), (int)scale2.size(), bCoarse, mbCheckOrientation, m12.data(), &nm));
        vMatchedPairs.clear();
        for (size_t i = 0; i < m12.size(); i++) if (m12[i] >= 0) vMatchedPairs.emplace_back(i, (size_t)m12[i]);     // :1203-1211
        return nm;
    }
    // search core of Fuse / SearchBySim3 / SearchByProjection(KF, Scw): see eorb_kf_radius_match
    void KeyFrameRadiusMatch(const FrameView& KF, const std::vector<uint8_t>& valid, const std::vector<float>& uv,
                             const std::vector<float>& radius, const std::vector<int>& level, const eorb_host::Mat8& mpDesc,
                             const std::vector<float>* invSigma2, std::vector<uint8_t>* taken, float acceptThr,
                             std::vector<int>& bestIdx, std::vector<int>& bestDist) {
        auto& c = eorb_host::thread_context();
        const int M = (int)valid.size();
        bestIdx.assign(M, -1); bestDist.assign(M, 256);
        c.check(eorb_kf_radius_match(c.get(), KF.kps->data(), KF.numAllKPts(), KF.desc->ptr(), KF.desc->cols, &KF.gb, M, valid.data(),
                                     uv.data(), radius.data(), level.data(), mpDesc.ptr(), invSigma2 ? invSigma2->data() : nullptr,
                                     invSigma2 ? (int)invSigma2->size() : 0, taken ? taken->data() : nullptr, acceptThr,
                                     bestIdx.data(), bestDist.data()));
    }
    // Fuse on a rectified-stereo KeyFrame (:1541-1553): uright = pKF->mvuRight, qUr[m] = u - bf * invz of map point m
    void FuseStereoMatch(const FrameView& KF, const std::vector<uint8_t>& valid, const std::vector<float>& uv,
                         const std::vector<float>& radius, const std::vector<int>& level, const eorb_host::Mat8& mpDesc,
                         const std::vector<float>& invSigma2, const std::vector<float>& uright, const std::vector<float>& qUr,
                         std::vector<int>& bestIdx, std::vector<int>& bestDist) {
        auto& c = eorb_host::thread_context();
        const int M = (int)valid.size();
        bestIdx.assign(M, -1); bestDist.assign(M, 256);
        c.check(eorb_kf_radius_match_stereo(c.get(), KF.kps->data(), KF.numAllKPts(), KF.desc->ptr(), KF.desc->cols, &KF.gb, M, valid.data(),
                                            uv.data(), radius.data(), level.data(), mpDesc.ptr(), invSigma2.data(), (int)invSigma2.size(),
                                            uright.data(), qUr.data(), bestIdx.data(), bestDist.data()));
    }
protected:
    float mfNNratio; bool mbCheckOrientation;
};

// ORBVocabulary (= DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) resident on the device
class ORBVocabulary {
public:
    // m_nodes flattened after loadFromTextFile: node 0 = root (see eorb_bow_set_vocabulary)
    ORBVocabulary(int L, const std::vector<int32_t>& childOff, const std::vector<int32_t>& childIds, const eorb_host::Mat8& nodeDesc,
                  const std::vector<int32_t>& wordId, const std::vector<double>& weight, int weighting = 0, int norm = 1)
        : weighting_(weighting), norm_(norm) {
        eorb_host::ContextPool::Voc v;
        v.L = L; v.childOff = childOff; v.childIds = childIds; v.wordId = wordId; v.weight = weight;
        v.nodeDesc.assign(nodeDesc.ptr(), nodeDesc.ptr() + (size_t)nodeDesc.rows * nodeDesc.cols);
        eorb_host::ContextPool::instance().set_vocabulary(std::move(v));                    // every context of the process
    }
    // transform(vCurrentDesc, mBowVec, mFeatVec, levelsup) (Frame::ComputeBoW)
    void transform(const eorb_host::Mat8& desc, std::vector<std::pair<uint32_t, double>>& bowVec, ORBmatcher::FeatureVector& featVec,
                   int levelsup = 4) const {
        auto& c = eorb_host::thread_context();
        const int n = desc.rows;
        std::vector<uint32_t> bw(n ? n : 1); std::vector<double> bv(n ? n : 1); int nw = 0, nn = 0;
        featVec.nodes.assign(n ? n : 1, 0); featVec.off.assign(n + 1, 0); featVec.idx.assign(n ? n : 1, 0);
        c.check(eorb_bow_transform(c.get(), desc.ptr(), n, desc.cols ? desc.cols : 32, levelsup, weighting_, norm_, bw.data(), bv.data(),
                                   &nw, featVec.nodes.data(), featVec.off.data(), featVec.idx.data(), &nn, nullptr, nullptr));
        bowVec.clear();
        for (int i = 0; i < nw; i++) bowVec.emplace_back(bw[i], bv[i]);
        featVec.nodes.resize(nn); featVec.off.resize(nn + 1); featVec.idx.resize(featVec.off[nn]);
    }
private:
    int weighting_, norm_;
};

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:349-423) for a batch of map points (CSR offsets into desc rows)
inline std::vector<int> ComputeDistinctiveDescriptors(const eorb_host::Mat8& desc, const std::vector<int32_t>& offsets) {
    auto& c = eorb_host::thread_context();
    std::vector<int> best(offsets.size() - 1, -1);
    c.check(eorb_distinctive_descriptors(c.get(), desc.ptr(), offsets.data(), (int)offsets.size() - 1, best.data()));
    return best;
}

}  // namespace ORB_SLAM3
