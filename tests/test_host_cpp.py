"""The C++ host mirror (eorb_slam_amd/host/eorb_host.hpp) must compile and link against libeorb_fe.so with a
plain g++ (no OpenCV, no HIP headers).  On a GPU box the same program runs and its output is checked."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "eorb_slam_amd/host/eorb_host.hpp"
#include <cstdio>
#include <thread>
int main(int argc, char** argv) {
    if (argc < 2) { std::puts("linked"); return 0; }          // link check only (no GPU touched)
    try {
        std::vector<eorb_host::EventData> ev(5000);
        for (size_t i = 0; i < ev.size(); i++) { ev[i].ts = 1e-6 * i; ev[i].x = (float)(20 + (i * 37) % 200); ev[i].y = (float)(20 + (i * 91) % 140); ev[i].p = i & 1; }
        eorb_host::Mat8 im8; eorb_host::Mat32f im32;
        EORB_SLAM::EvImConverter::ev2im_gauss(ev, 240, 180, 1.0f, false, true, im8, im32);
        ORB_SLAM3::ORBxParams p; p.nfeatures = 400; p.scaleFactor = 1.0f; p.nlevels = 1; p.iniThFAST = 0; p.minThFAST = 0; p.edgeTh = 9; p.imWidth = 240; p.imHeight = 180;
        ORB_SLAM3::ORBextractor ex(p);
        std::vector<eorb_host::KeyPoint> kps; std::vector<int> lap{0, 1000};
        int mono = ex(im8, kps, lap);
        eorb_host::Mat8 empty;
        int m1 = ex(empty, kps, lap);
        std::printf("mono=%d empty=%d\n", mono, m1);
        // loader on the device: text -> raw events -> rectified EventData == the fused raw image path
        std::vector<float> mx(240 * 180), my(240 * 180);
        for (int y = 0; y < 180; y++) for (int x = 0; x < 240; x++) { mx[y * 240 + x] = x + 0.25f * (y % 3) - 0.4f; my[y * 240 + x] = y + 0.125f * (x % 5); }
        EORB_SLAM::EventDataStore::setUndistortMaps(mx, my, 240, 180, true);
        std::string text = "# ts x y p\n";
        for (int i = 0; i < 3000; i++) { char b[64]; std::snprintf(b, sizeof b, "%d.%06d %d %d %d\n", i / 1000, (i * 37) % 1000000, (i * 53) % 240, (i * 29) % 180, i & 1); text += b; }
        auto raw = EORB_SLAM::EventDataStore::parseText(text);
        auto evs = EORB_SLAM::EventDataStore::rectify(raw, 240, 180, 1.0);
        eorb_host::Mat8 a8, b8; eorb_host::Mat32f a32, b32;
        EORB_SLAM::EvImConverter::ev2im_gauss(evs, 240, 180, 1.0f, true, false, a8, a32);
        EORB_SLAM::EvImConverter::ev2im_gauss_raw(raw, 240, 180, 1.0f, true, false, b8, b32);
        bool same = raw.size() == 3000 && evs.size() < raw.size() && evs.size() > 2000;
        for (int i = 0; i < 240 * 180 && same; i++) same = a32.ptr()[i] == b32.ptr()[i];
        // one map point observed 5 times: the distinctive descriptor is one of the rows
        eorb_host::Mat8 d(5, 32); for (int i = 0; i < 5 * 32; i++) d.ptr()[i] = (unsigned char)(i * 7 + (i / 32 == 3 ? 1 : 0));
        auto best = ORB_SLAM3::ComputeDistinctiveDescriptors(d, std::vector<int32_t>{0, 5});
        std::printf("raw=%zu kept=%zu same=%d best=%d\n", raw.size(), evs.size(), (int)same, best[0]);
        // transient threads (EvImBuilder.cpp:1165-1193 starts four per motion-compensated image) borrow warm contexts: three rounds
        // of four threads must not create more than four contexts beyond the main thread's, and they see the maps set above
        bool pool_ok = true;
        for (int round = 0; round < 3; round++) {
            std::vector<std::thread> th; std::vector<int> okv(4, 0);
            for (int t = 0; t < 4; t++) th.emplace_back([&, t] {
                eorb_host::Mat8 c8; eorb_host::Mat32f c32;
                EORB_SLAM::EvImConverter::ev2im_gauss_raw(raw, 240, 180, 1.0f, true, false, c8, c32);
                bool s2 = true;
                for (int i = 0; i < 240 * 180 && s2; i++) s2 = c32.ptr()[i] == b32.ptr()[i];
                okv[t] = s2;
            });
            for (auto& x : th) x.join();
            for (int v : okv) pool_ok = pool_ok && v;
        }
        // the L1 builder's chunk loop over the one-call seams: an INIT chunk, then tracking chunks until the window rule dispatches; the
        // INIT frame and the first tracked frame equal the separate seams (ev2im_gauss + extractor / LK are checked in Python)
        bool chain_ok = true;
        {
            EORB_SLAM::EvImBuilder::Params bp;
            EORB_SLAM::EvImBuilder b(bp);
            std::vector<eorb_host::EventData> chunk(2000);
            int ndisp = 0, ninit = 0, ntrack = 0;
            for (int k = 0; k < 12; k++) {
                for (size_t i = 0; i < chunk.size(); i++) {                  // four edges drifting to the right, 2 px per chunk
                    const unsigned h = (unsigned)(k * 2000 + i) * 2654435761u;
                    const int e = (h >> 8) % 4, t = (h >> 12) % 120;
                    chunk[i].ts = 1e-6 * (k * 2000.0 + i); chunk[i].x = (float)(30 + 45 * e + 2 * k + (t % 3)); chunk[i].y = (float)(20 + t); chunk[i].p = (h >> 30) & 1;
                }
                auto r = b.Track(chunk);
                if (k == 0) {
                    eorb_host::Mat8 i8; eorb_host::Mat32f i32;
                    EORB_SLAM::EvImConverter::ev2im_gauss(chunk, 240, 180, 1.0f, false, true, i8, i32);
                    ORB_SLAM3::ORBxParams q; q.nfeatures = 400; q.scaleFactor = 1.0f; q.nlevels = 1; q.iniThFAST = 0; q.minThFAST = 0; q.edgeTh = 9; q.imWidth = 240; q.imHeight = 180;
                    ORB_SLAM3::ORBextractor ex2(q);
                    std::vector<eorb_host::KeyPoint> k2;
                    ex2(i8, k2, std::vector<int>{0, 1000});
                    chain_ok = chain_ok && r.state == EORB_SLAM::EvImBuilder::INIT && k2.size() == r.kps.size() &&
                               (k2.empty() || std::memcmp(k2.data(), r.kps.data(), k2.size() * sizeof(eorb_host::KeyPoint)) == 0);
                }
                ninit += r.state == EORB_SLAM::EvImBuilder::INIT; ntrack += r.state == EORB_SLAM::EvImBuilder::TRACKING;
                if (r.dispatched) { ndisp++; chain_ok = chain_ok && r.winner == 2 && r.focus[2] > 0 && r.focus[0] == -1.f && r.window >= 4000 && r.overlap == r.window / 2; }
            }
            std::printf("chain: init %d tracking %d dispatches %d ok=%d\n", ninit, ntrack, ndisp, (int)chain_ok);
            chain_ok = chain_ok && ndisp >= 1 && ntrack >= 2;
        }
        // the stereo constructor's path: a textured image and the same image moved by 5 pixels -> disparities of 5
        bool stereo_ok = true;
        {
            const int W = 240, H = 180;
            eorb_host::Mat8 L(H, W), R(H, W);
            auto tex = [](int x, int y) { unsigned h = (unsigned)(x / 3) * 73856093u ^ (unsigned)(y / 3) * 19349663u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; return (unsigned char)(h & 0xff); };
            for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { L.ptr()[y * W + x] = tex(x, y); R.ptr()[y * W + x] = tex(x + 5, y); }
            ORB_SLAM3::ORBxParams q; q.nfeatures = 1000; q.scaleFactor = 1.2f; q.nlevels = 4; q.iniThFAST = 20; q.minThFAST = 7; q.edgeTh = 19; q.imWidth = W; q.imHeight = H;
            ORB_SLAM3::ORBextractor exs(q);
            std::vector<eorb_host::KeyPoint> kl, kr; eorb_host::Mat8 dl, dr; std::vector<float> ur, dp;
            const int nm = exs.ExtractStereo(L, R, 0.11f, 40.0f, kl, dl, kr, dr, ur, dp);
            int good = 0, matched = 0;
            for (size_t i = 0; i < kl.size(); i++) if (ur[i] > 0) { matched++; good += std::fabs(kl[i].x - ur[i] - 5.0f) < 1.5f && dp[i] == 40.0f / (kl[i].x - ur[i]); }
            std::printf("stereo: %zu / %zu keypoints, %d correlated, %d kept, %d at the planted disparity\n", kl.size(), kr.size(), nm, matched, good);
            stereo_ok = kl.size() > 100 && matched > 50 && good * 10 >= matched * 9 && nm >= matched;
        }
        const size_t made = eorb_host::ContextPool::instance().created();
        std::printf("pool contexts=%zu ok=%d\n", made, (int)pool_ok);
        return (mono == 0 && m1 == -1 && same && best[0] >= 0 && best[0] < 5 && pool_ok && made <= 6 && chain_ok && stereo_ok) ? 0 : 1;
    } catch (const eorb_host::Error& e) { std::printf("error %d: %s\n", e.code, e.what()); return 2; }
}
'''


def _build(tmp):
    from eorb_slam_amd import _lib
    lib = _lib.build()
    src = os.path.join(tmp, "host_check.cpp"); exe = os.path.join(tmp, "host_check")
    open(src, "w").write(SRC)
    libdir = os.path.dirname(lib)
    p = subprocess.run(["g++", "-std=c++14", "-Wall", "-I", ROOT, src, "-o", exe, "-L", libdir, "-leorb_fe",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_host_mirror_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path))
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "linked" in out.stdout


@pytest.mark.gpu
def test_host_mirror_runs(tmp_path):
    exe = _build(str(tmp_path))
    out = subprocess.run([exe, "run"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_latency_tool_compiles(tmp_path):
    """tools/latency.cpp (per-slice latency through the bare C ABI) must keep compiling and linking."""
    from eorb_slam_amd import _lib
    libdir = os.path.dirname(_lib.build())
    exe = os.path.join(str(tmp_path), "latency")
    p = subprocess.run(["g++", "-O2", "-std=c++14", "-Wall", "-I", ROOT, os.path.join(ROOT, "tools", "latency.cpp"), "-o", exe, "-L", libdir,
                        "-leorb_fe", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
