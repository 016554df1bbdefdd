// The context pool of eorb_slam_amd/host/eorb_host.hpp under transient threads, as src/Event/EvImBuilder.cpp:1165-1193 uses the
// converters (four std::threads per motion-compensated image) while another thread replaces the calibrator's maps.  Linked against
// tests/host/eorb_stub.c and built with -fsanitize=thread by tests/test_oracle_hygiene.py.
#include "../../eorb_slam_amd/host/eorb_host.hpp"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
extern "C" void eorb_stub_fail_once(void);
extern "C" int eorb_stub_inconsistent(void);

int main()
{
    auto& pool = eorb_host::ContextPool::instance();
    std::atomic<bool> stop{false};
    std::thread setter([&] {
        for (int it = 0; !stop.load(); it++) {
            const int LW = 8 + it % 5, LH = 4;
            std::vector<float> mx((size_t)LW * LH, (float)LW), my((size_t)LW * LH, 0.f);
            pool.set_maps(mx, my, LW, LH, true);
            std::this_thread::yield();
        }
    });
    int failures_seen = 0;
    for (int round = 0; round < 50; round++) {
        if (round == 20) eorb_stub_fail_once();
        std::vector<std::thread> th;
        std::atomic<int> threw{0};
        for (int t = 0; t < 4; t++) th.emplace_back([&] {
            try {
                for (int k = 0; k < 20; k++) { auto& c = eorb_host::thread_context(); (void)c; std::this_thread::sleep_for(std::chrono::microseconds(50)); }
            } catch (const eorb_host::Error&) { threw++; }
        });
        for (auto& x : th) x.join();
        failures_seen += threw.load();
    }
    stop = true; setter.join();
    const size_t made = pool.created();
    std::printf("created=%zu failures=%d inconsistent=%d\n", made, failures_seen, eorb_stub_inconsistent());
    // four transient threads at a time (+ the context stranded-then-returned by the injected failure must be reusable): never more than 5
    return (made <= 5 && failures_seen == 1 && !eorb_stub_inconsistent()) ? 0 : 1;
}
