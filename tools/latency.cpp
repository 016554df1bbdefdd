// Latency of the per-slice / per-frame seams through the C ABI, without a binding in between (bench.py's figures include ctypes / numpy):
//   W1  eorb_ev2im_gauss_raw (2 000 sensor events -> u8 image)  ->  eorb_orb_extract (FAST detection of 400 points, 1 level)
//   W3  eorb_orb_extract (240x180, ORB-1000) -> mixed eorb_search_for_initialization -> eorb_search_by_projection_last, one frame per call,
//       as Tracking::GrabImage* / Frame construction call the seams (src/Tracking.cc:1418-1427, :1816-1866; src/Frame.cc:467-482)
//   W4  eorb_orb_extract (346x260, 8 levels, 2 000 features) -> eorb_hamming_bf_knn2 2000 x 2000
// build:  g++ -O2 -std=c++14 -I. tools/latency.cpp -o /tmp/latency -Leorb_slam_amd/csrc -leorb_fe -Wl,-rpath,$PWD/eorb_slam_amd/csrc -Wl,-rpath,/opt/rocm/lib
#include "include/eorb_fe.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pct(const char* name, std::vector<double> v) { std::sort(v.begin(), v.end()); std::printf("%-28s p50 %.4f ms  p95 %.4f ms\n", name, 1e3 * v[v.size() / 2], 1e3 * v[v.size() * 95 / 100]); }
static unsigned g_s = 777u;
static unsigned rnd() { g_s = g_s * 1664525u + 1013904223u; return g_s >> 8; }
// a textured frame (smooth random field + hard-edged rectangles), shifted by (dx, dy): corners on every pyramid level
static std::vector<uint8_t> texture(int W, int H, int dx, int dy, unsigned seed)
{
    g_s = seed;
    std::vector<float> f((size_t)W * H, 0.f);
    for (int o = 0; o < 6; o++) {
        const float fx = 0.02f * (1 << (o / 2)) * (1.f + (rnd() % 100) / 100.f), fy = 0.02f * (1 << (o / 2)) * (1.f + (rnd() % 100) / 100.f);
        const float ph = (rnd() % 628) / 100.f, am = 1.f / (1 + o / 2);
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) f[(size_t)y * W + x] += am * std::sin(fx * (x + dx) + fy * (y + dy) * 1.3f + ph) * std::cos(fy * (x + dx) * 0.7f - fx * (y + dy));
    }
    for (int r = 0; r < 160; r++) {
        const int w = 5 + rnd() % 26, h = 5 + rnd() % 26, x0 = (int)(rnd() % (W + 40)) - 20 - dx, y0 = (int)(rnd() % (H + 40)) - 20 - dy;
        const float v = ((int)(rnd() % 200) - 100) / 100.f;
        for (int y = y0 < 0 ? 0 : y0; y < y0 + h && y < H; y++) for (int x = x0 < 0 ? 0 : x0; x < x0 + w && x < W; x++) f[(size_t)y * W + x] += v;
    }
    float lo = 1e9f, hi = -1e9f;
    for (float v : f) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    std::vector<uint8_t> img((size_t)W * H);
    for (size_t i = 0; i < img.size(); i++) img[i] = (uint8_t)std::lround(255.f * (f[i] - lo) / (hi - lo));
    return img;
}

static int run_w3(eorb_ctx* c)
{
    const int W = 240, H = 180, NFR = 16, REP = 300, NA = 500;
    eorb_orb_params p{}; p.nfeatures = 1000; p.scaleFactor = 1.2f; p.nlevels = 4; p.iniThFAST = 10; p.minThFAST = 0; p.edgeTh = 19;
    if (eorb_orb_configure(c, &p, W, H)) { std::printf("configure: %s\n", eorb_last_error(c)); return 1; }
    const int cap = eorb_orb_max_keypoints(c), CAP = cap + NA;
    std::vector<std::vector<uint8_t>> frames;
    for (int i = 0; i < NFR; i++) frames.push_back(texture(W, H, (2 * i) % 11, i % 7, 4242u));
    struct Fr { std::vector<eorb_keypoint> k; std::vector<uint8_t> d, o; int n = 0; };
    Fr prev, cur; prev.k.resize(CAP); prev.d.resize((size_t)CAP * 61); prev.o.resize(CAP); cur = prev;
    std::vector<uint8_t> desc((size_t)cap * 32), oob(cap), valid(CAP), mpd((size_t)CAP * 32), mpo(CAP, 1);
    std::vector<float> pm((size_t)CAP * 2), uv((size_t)CAP * 2), ls(CAP);
    std::vector<int32_t> m12(CAP), curmp(CAP);
    const eorb_grid_bounds gb{0.f, 0.f, (float)W, (float)H, 64.f / W, 48.f / H};
    std::vector<double> te, ti, tp, tt;
    int n1 = 0, n2 = 0, nk = 0;
    for (int r = 0; r < REP + 10; r++) {
        const double t0 = now();
        int n = 0, mono = 0;
        if (eorb_orb_extract(c, frames[r % NFR].data(), W, H, W, 0, 1000, 1, cur.k.data(), desc.data(), oob.data(), cap, &n, &mono) < 0) { std::printf("extract: %s\n", eorb_last_error(c)); return 1; }
        // MixedFrame stand-in: the ORB rows + 500 AKAZE-like rows (61 bytes, the first 32 compared), type flags
        for (int i = 0; i < n; i++) { std::memcpy(&cur.d[(size_t)i * 61], &desc[(size_t)i * 32], 32); cur.o[i] = 1; }
        g_s = 99u + r % NFR;
        for (int i = n; i < n + NA; i++) {
            cur.k[i].x = (float)(rnd() % (W * 16)) / 16.f; cur.k[i].y = (float)(rnd() % (H * 16)) / 16.f; cur.k[i].size = 10.f; cur.k[i].angle = (float)(rnd() % 360);
            cur.k[i].response = 0.01f; cur.k[i].octave = 0; cur.k[i].class_id = (int)(rnd() % 3);
            for (int b = 0; b < 61; b++) cur.d[(size_t)i * 61 + b] = (uint8_t)rnd();
            cur.o[i] = 0;
        }
        cur.n = n + NA;
        const double t1 = now();
        double t2 = t1, t3 = t1;
        if (prev.n) {
            for (int i = 0; i < prev.n; i++) { pm[2 * i] = prev.k[i].x; pm[2 * i + 1] = prev.k[i].y; }
            if (eorb_search_for_initialization(c, prev.k.data(), prev.n, prev.d.data(), 61, prev.o.data(), cur.k.data(), cur.n, cur.d.data(), 61, cur.o.data(),
                                               &gb, pm.data(), m12.data(), 100, 0.9f, 1, &n1)) { std::printf("init: %s\n", eorb_last_error(c)); return 1; }
            t2 = now();
            for (int i = 0; i < prev.n; i++) {
                valid[i] = (i % 5) != 0; uv[2 * i] = prev.k[i].x + 0.5f; uv[2 * i + 1] = prev.k[i].y - 0.5f;
                std::memcpy(&mpd[(size_t)i * 32], &prev.d[(size_t)i * 61], 32);
                ls[i] = prev.o[i] ? std::pow(1.2f, (float)prev.k[i].octave) : std::pow(1.26f, (float)prev.k[i].class_id);
            }
            for (int i = 0; i < cur.n; i++) curmp[i] = -1;
            if (eorb_search_by_projection_last(c, cur.k.data(), cur.n, cur.d.data(), 61, cur.o.data(), prev.k.data(), prev.n, prev.o.data(), valid.data(), uv.data(),
                                               mpd.data(), mpo.data(), ls.data(), &gb, curmp.data(), 15.0f, 0, 1, &n2)) { std::printf("proj: %s\n", eorb_last_error(c)); return 1; }
            t3 = now();
        }
        if (r >= 10) { te.push_back(t1 - t0); ti.push_back(t2 - t1); tp.push_back(t3 - t2); tt.push_back(t3 - t0); }
        std::swap(prev, cur); nk = n;
    }
    std::printf("W3: keypoints %d (+%d AKAZE-like), matches init %d, proj %d\n", nk, NA, n1, n2);
    pct("W3 extract (+ host mixing)", te); pct("W3 SearchForInitialization", ti); pct("W3 SearchByProjection(last)", tp); pct("W3 frame", tt);
    return 0;
}

static int run_w4(eorb_ctx* c)
{
    const int W = 346, H = 260, NFR = 16, REP = 300, NQ = 2000;
    eorb_orb_params p{}; p.nfeatures = 2000; p.scaleFactor = 1.2f; p.nlevels = 8; p.iniThFAST = 10; p.minThFAST = 0; p.edgeTh = 15;
    if (eorb_orb_configure(c, &p, W, H)) { std::printf("configure: %s\n", eorb_last_error(c)); return 1; }
    const int cap = eorb_orb_max_keypoints(c);
    std::vector<std::vector<uint8_t>> frames;
    for (int i = 0; i < NFR; i++) frames.push_back(texture(W, H, (3 * i) % 13, i % 5, 5151u));
    std::vector<eorb_keypoint> kps(cap); std::vector<uint8_t> desc((size_t)cap * 32), oob(cap), q((size_t)NQ * 32), t((size_t)NQ * 32);
    g_s = 5u;
    for (auto& b : t) b = (uint8_t)rnd();
    for (int i = 0; i < NQ; i++) { std::memcpy(&q[(size_t)i * 32], &t[(size_t)((i * 7) % NQ) * 32], 32); for (int k = 0; k < (int)(rnd() % 40); k++) q[(size_t)i * 32 + rnd() % 32] ^= (uint8_t)(1u << (rnd() % 8)); }
    std::vector<int32_t> idx2((size_t)NQ * 2), d2((size_t)NQ * 2);
    std::vector<double> te, tb, tt;
    int n = 0, mono = 0;
    for (int r = 0; r < REP + 10; r++) {
        const double t0 = now();
        if (eorb_orb_extract(c, frames[r % NFR].data(), W, H, W, 0, 1000, 1, kps.data(), desc.data(), oob.data(), cap, &n, &mono) < 0) { std::printf("extract: %s\n", eorb_last_error(c)); return 1; }
        const double t1 = now();
        if (eorb_hamming_bf_knn2(c, q.data(), NQ, t.data(), NQ, idx2.data(), d2.data())) { std::printf("bf: %s\n", eorb_last_error(c)); return 1; }
        const double t2 = now();
        if (r >= 10) { te.push_back(t1 - t0); tb.push_back(t2 - t1); tt.push_back(t2 - t0); }
    }
    std::printf("W4: keypoints %d\n", n);
    pct("W4 extract", te); pct("W4 BF 2-NN 2000x2000", tb); pct("W4 frame", tt);
    return 0;
}

int main()
{
    const int W = 240, H = 180, N = 2000, REP = 500;
    eorb_ctx* c = nullptr;
    if (eorb_create(0, nullptr, &c)) { std::puts("no device"); return 2; }
    std::vector<float> mx(W * H), my(W * H);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { mx[y * W + x] = x + 0.3f * ((y % 7) / 7.0f) - 0.1f; my[y * W + x] = y + 0.25f * ((x % 5) / 5.0f); }
    eorb_set_undistort_maps(c, mx.data(), my.data(), W, H, 1);
    eorb_orb_params p{}; p.nfeatures = 400; p.scaleFactor = 1.0f; p.nlevels = 1; p.iniThFAST = 0; p.minThFAST = 0; p.edgeTh = 9;
    if (eorb_orb_configure(c, &p, W, H)) { std::printf("configure: %s\n", eorb_last_error(c)); return 1; }
    const int cap = eorb_orb_max_keypoints(c);
    std::vector<eorb_raw_event> ev(N);
    unsigned s = 12345;
    for (int i = 0; i < N; i++) {       // three moving edges
        s = s * 1664525u + 1013904223u;
        const int e = (s >> 8) % 3, t = (s >> 12) % 160;
        ev[i].x = (uint16_t)(20 + 60 * e + (t % 50)); ev[i].y = (uint16_t)(10 + t); ev[i].p = (s >> 30) & 1; ev[i].t = 1e-6 * i;
    }
    std::vector<uint8_t> img(W * H), oob(cap);
    std::vector<eorb_keypoint> kps(cap);
    float mm[2]; int n = 0, mono = 0;
    std::vector<double> a, b, t, one;
    for (int r = 0; r < REP + 20; r++) {
        const double t0 = now();
        if (eorb_ev2im_gauss_raw(c, ev.data(), N, W, H, 1.0f, 0, 1, nullptr, img.data(), mm)) { std::printf("ev2im: %s\n", eorb_last_error(c)); return 1; }
        const double t1 = now();
        if (eorb_orb_extract(c, img.data(), W, H, W, 0, 1000, 0, kps.data(), nullptr, oob.data(), cap, &n, &mono) < 0) { std::printf("extract: %s\n", eorb_last_error(c)); return 1; }
        const double t2 = now();
        eorb_ev2im_gauss_raw(c, ev.data(), 1, W, H, 1.0f, 0, 1, nullptr, img.data(), mm);
        const double t3 = now();
        if (r >= 20) { a.push_back(t1 - t0); b.push_back(t2 - t1); t.push_back(t2 - t0); one.push_back(t3 - t2); }
    }
    std::printf("keypoints %d\n", n);
    pct("eorb_ev2im_gauss_raw (2000)", a); pct("eorb_orb_extract (detect)", b); pct("both", t); pct("eorb_ev2im_gauss_raw (1 ev)", one);
    // the same chunk through the one-call seams (EvImBuilder::Track's per-chunk path): events in -> keypoints out / tracked points out
    {
        std::vector<eorb_event> fev(N);
        for (int i = 0; i < N; i++) { fev[i].ts = ev[i].t; fev[i].x = mx[ev[i].y * W + ev[i].x]; fev[i].y = my[ev[i].y * W + ev[i].x]; fev[i].p = ev[i].p != 0; }
        std::vector<eorb_raw_event> ev2(ev);
        for (int i = 0; i < N; i++) ev2[i].x = (uint16_t)(ev2[i].x + 1);                  // the next chunk: the edges moved by a pixel
        eorb_klt_params klt{23, 1, 10, 0.03, 1e-4f};
        std::vector<double> xr, xf, tr;
        std::vector<float> pts((size_t)cap * 2), err(cap); std::vector<uint8_t> st(cap);
        int nk = 0;
        for (int r = 0; r < REP + 20; r++) {
            const double t0 = now();
            if (eorb_ev_slice_extract(c, nullptr, ev.data(), N, 1.0f, 0, 1000, 0, kps.data(), nullptr, nullptr, cap, &nk, &mono, nullptr)) { std::printf("slice_extract: %s\n", eorb_last_error(c)); return 1; }
            const double t1 = now();
            for (int i = 0; i < nk; i++) { pts[2 * i] = kps[i].x; pts[2 * i + 1] = kps[i].y; }
            const double t2 = now();
            if (eorb_ev_slice_track(c, nullptr, ev2.data(), N, 1.0f, &klt, pts.data(), st.data(), err.data(), nk, nullptr)) { std::printf("slice_track: %s\n", eorb_last_error(c)); return 1; }
            const double t3 = now();
            if (eorb_ev_slice_extract(c, fev.data(), nullptr, N, 1.0f, 0, 1000, 0, kps.data(), nullptr, nullptr, cap, &nk, &mono, nullptr)) { std::printf("slice_extract: %s\n", eorb_last_error(c)); return 1; }
            const double t4 = now();
            if (r >= 20) { xr.push_back(t1 - t0); tr.push_back(t3 - t2); xf.push_back(t4 - t3); }
        }
        int ok = 0; for (int i = 0; i < nk; i++) ok += st[i];
        std::printf("one-call seams: %d keypoints, %d tracked\n", nk, ok);
        pct("eorb_ev_slice_extract (raw)", xr); pct("eorb_ev_slice_extract (float)", xf); pct("eorb_ev_slice_track (raw)", tr);
    }
    if (run_w3(c) || run_w4(c)) return 1;
    eorb_destroy(c);
    return 0;
}
