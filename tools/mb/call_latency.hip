// call_latency.hip -- the fixed cost of a host-buffer call: ways to get a few KB to the GPU, run a kernel chain and get a few KB back.
// hipcc -O3 --offload-arch=gfx950 tools/mb/call_latency.hip -o call_latency && ./call_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void work(const uint32_t* in, uint32_t* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 3u + 1u;
}
// last kernel of a chain: copies the results to host memory and raises the flag (system scope)
__global__ void publish(const uint32_t* res, uint32_t* host_out, int n, volatile uint32_t* flag, uint32_t seq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) host_out[i] = res[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __shared__ int dummy;
        // one block only in this test (n <= 1024)
        __hip_atomic_store((uint32_t*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        (void)dummy;
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const int n = 1024, iters = 3000, chain = argc > 1 ? atoi(argv[1]) : 4;
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t *hin, *hout, *din, *dmid, *dout; volatile uint32_t* flag;
    CHECK(hipHostMalloc((void**)&hin, n * 4, hipHostMallocDefault)); CHECK(hipHostMalloc((void**)&hout, n * 4, hipHostMallocDefault));
    CHECK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
    CHECK(hipMalloc(&din, n * 4)); CHECK(hipMalloc(&dmid, n * 4)); CHECK(hipMalloc(&dout, n * 4));
    hipEvent_t ev; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int i = 0; i < n; i++) hin[i] = i;
    *flag = 0;
    auto run = [&](const char* name, auto&& body) {
        std::vector<double> t(iters);
        for (int it = 0; it < iters + 200; it++) {
            const double t0 = now();
            body((uint32_t)(it + 1));
            const double t1 = now();
            if (it >= 200) t[it - 200] = (t1 - t0) * 1e6;
        }
        std::sort(t.begin(), t.end());
        printf("%-78s p50 %7.1f us  p95 %7.1f us\n", name, t[iters / 2], t[iters * 95 / 100]);
    };
    auto kernels = [&](const uint32_t* src) {
        const uint32_t* a = src;
        for (int k = 0; k < chain; k++) { uint32_t* o = (k & 1) ? dout : dmid; work<<<1, 1024, 0, s>>>(a, o, n); a = o; }
        return a;
    };
    printf("chain of %d kernels, %d bytes in and out\n", chain, n * 4);
    run("A  H2D copy, kernels, D2H copy, event record + hipEventSynchronize", [&](uint32_t) {
        CHECK(hipMemcpyAsync(din, hin, n * 4, hipMemcpyHostToDevice, s)); const uint32_t* r = kernels(din);
        CHECK(hipMemcpyAsync(hout, r, n * 4, hipMemcpyDeviceToHost, s)); CHECK(hipEventRecord(ev, s)); CHECK(hipEventSynchronize(ev)); });
    run("B  same, spinning on hipEventQuery", [&](uint32_t) {
        CHECK(hipMemcpyAsync(din, hin, n * 4, hipMemcpyHostToDevice, s)); const uint32_t* r = kernels(din);
        CHECK(hipMemcpyAsync(hout, r, n * 4, hipMemcpyDeviceToHost, s)); CHECK(hipEventRecord(ev, s)); while (hipEventQuery(ev) == hipErrorNotReady) {} });
    run("C  same, hipStreamSynchronize", [&](uint32_t) {
        CHECK(hipMemcpyAsync(din, hin, n * 4, hipMemcpyHostToDevice, s)); const uint32_t* r = kernels(din);
        CHECK(hipMemcpyAsync(hout, r, n * 4, hipMemcpyDeviceToHost, s)); CHECK(hipStreamSynchronize(s)); });
    run("D  H2D copy, kernels, publish kernel (writes host memory + flag), host spins on the flag", [&](uint32_t seq) {
        CHECK(hipMemcpyAsync(din, hin, n * 4, hipMemcpyHostToDevice, s)); const uint32_t* r = kernels(din);
        publish<<<1, 1024, 0, s>>>(r, hout, n, flag, seq); while (*flag != seq) {} });
    run("E  first kernel reads pinned host memory, publish kernel + flag spin (no copies)", [&](uint32_t seq) {
        const uint32_t* r = kernels(hin);
        publish<<<1, 1024, 0, s>>>(r, hout, n, flag, seq); while (*flag != seq) {} });
    run("F  first kernel reads pinned host memory, D2H copy + hipEventSynchronize", [&](uint32_t) {
        const uint32_t* r = kernels(hin);
        CHECK(hipMemcpyAsync(hout, r, n * 4, hipMemcpyDeviceToHost, s)); CHECK(hipEventRecord(ev, s)); CHECK(hipEventSynchronize(ev)); });
    run("G  first kernel reads pinned host memory, publish kernel, hipStreamSynchronize", [&](uint32_t seq) {
        const uint32_t* r = kernels(hin);
        publish<<<1, 1024, 0, s>>>(r, hout, n, flag, seq); CHECK(hipStreamSynchronize(s)); });
    run("H  kernels only + hipStreamSynchronize (no data)", [&](uint32_t) { kernels(din); CHECK(hipStreamSynchronize(s)); });
    run("I  kernels only + flag spin", [&](uint32_t seq) { const uint32_t* r = kernels(din); publish<<<1, 1024, 0, s>>>(r, hout, n, flag, seq); while (*flag != seq) {} });
    for (int i = 0; i < n; i++) if (hout[i] == 0xdeadbeef) printf("?");
    return 0;
}
