// Latency of the per-slice seams through the C ABI, without a binding in between (bench.py's latency figures include ctypes / numpy):
//   eorb_ev2im_gauss_raw (2 000 sensor events -> u8 image)  ->  eorb_orb_extract (FAST detection of 400 points, 1 level)
// build:  g++ -O2 -std=c++14 -I. tools/latency.cpp -o /tmp/latency -Leorb_slam_amd/csrc -leorb_fe -Wl,-rpath,$PWD/eorb_slam_amd/csrc -Wl,-rpath,/opt/rocm/lib
#include "include/eorb_fe.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pct(const char* name, std::vector<double> v) { std::sort(v.begin(), v.end()); std::printf("%-28s p50 %.4f ms  p95 %.4f ms\n", name, 1e3 * v[v.size() / 2], 1e3 * v[v.size() * 95 / 100]); }
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
    eorb_destroy(c);
    return 0;
}
