// Config 1 as the reference runs it, from C++: EvImBuilder::Track (src/Event/EvImBuilder.cpp:1300-1515) through the host mirror
// eorb_slam_amd/host/eorb_host.hpp -- per 2 000-event chunk ev2im_gauss -> FAST 400 (INIT) or LK against the reference frame (TRACKING),
// every dispatch the four-way reconstruction contest and the detection on the winner -- with nothing but the C ABI underneath
// (bench.py --workload w1full runs the same chain through ctypes and numpy).  Prints the time of a Track() call by kind and chunks/s.
// build:  g++ -O2 -std=c++14 -I. tools/chain_latency.cpp -o build/chain_latency -Leorb_slam_amd/csrc -leorb_fe -lpthread -Wl,-rpath,$PWD/eorb_slam_amd/csrc -Wl,-rpath,/opt/rocm/lib
#include "eorb_slam_amd/host/eorb_host.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Poses { EORB_SLAM::EvImBuilder::MciPoses p; eorb_se3_motion dp, ba; float se2[3]; eorb_camera cam; };
static const EORB_SLAM::EvImBuilder::MciPoses* give_poses(const std::vector<eorb_host::EventData>&, void* user) { return &((Poses*)user)->p; }

int main()
{
    const int W = 240, H = 180, NCH = 400, N = 2000;
    Poses P{};
    P.cam = eorb_camera{0, 199.092366542f, 198.82882047f, 132.192071378f, 110.712660011f, {0, 0, 0, 0}, 1e-6f};
    P.dp = eorb_se3_motion{0.004, {0.1, -0.3, 0.95}, {0.002, -0.001, 0.0005}, 1.7f};
    P.ba = eorb_se3_motion{0.003, {0.0, 0.2, 0.98}, {0.0015, 0.0008, -0.0004}, 1.6f};
    P.se2[0] = 0.002f; P.se2[1] = 1.5f; P.se2[2] = -0.4f;
    P.p.dp = &P.dp; P.p.ba = &P.ba; P.p.se2 = P.se2; P.p.nse2 = 3; P.p.cam = &P.cam;
    EORB_SLAM::EvImBuilder::Params bp;
    EORB_SLAM::EvImBuilder b(bp);
    std::vector<eorb_host::EventData> chunk(N);
    std::vector<double> t_init, t_track, t_disp;
    double total = 0;
    int nkp = 0;
    for (int k = 0; k < NCH + 20; k++) {
        for (int i = 0; i < N; i++) {                                   // 160 small blobs drifting to the right by 0.4 px per chunk
            const unsigned h = (unsigned)(k * N + i) * 2654435761u;
            const unsigned j = (h >> 7) % 160u, pj = j * 2246822519u + 374761393u;
            const float bx = 14.f + (float)((pj >> 4) % 190u), by = 14.f + (float)((pj >> 14) % 150u);
            chunk[i].ts = 1e-6 * (k * (double)N + i);
            chunk[i].x = bx + 0.4f * (float)(k % 40) + 0.5f * (float)((h >> 20) & 3) - 0.75f;
            chunk[i].y = by + 0.5f * (float)((h >> 24) & 3) - 0.75f;
            chunk[i].p = (h >> 30) & 1;
        }
        const double t0 = now();
        auto r = b.Track(chunk, give_poses, &P);
        const double dt = (now() - t0) * 1e3;
        if (k < 20) continue;                                           // warm-up
        total += dt;
        if (r.dispatched) t_disp.push_back(dt);
        else if (r.state == EORB_SLAM::EvImBuilder::INIT) { t_init.push_back(dt); nkp = (int)r.kps.size(); }
        else t_track.push_back(dt);
    }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::printf("EvImBuilder::Track from C++, %d chunks of %d events (%d keypoints per INIT frame):\n", NCH, N, nkp);
    std::printf("  INIT chunk (image + FAST 400)                  %4zu calls  p50 %.4f ms\n", t_init.size(), med(t_init));
    std::printf("  TRACKING chunk (image + LK)                    %4zu calls  p50 %.4f ms\n", t_track.size(), med(t_track));
    std::printf("  dispatching chunk (+ contest + detection)      %4zu calls  p50 %.4f ms\n", t_disp.size(), med(t_disp));
    std::printf("  all: %.4f ms per chunk = %.0f chunks/s\n", total / NCH, NCH / (total * 1e-3));
    return 0;
}
