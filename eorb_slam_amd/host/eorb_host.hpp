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
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/eorb_fe.h"

namespace eorb_host {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// one context per calling thread (the ABI's threading contract)
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
private:
    eorb_ctx* h_ = nullptr;
};

inline Context& thread_context() { thread_local Context c; return c; }

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
protected:
    float mfNNratio; bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3
