// clock_probe.hip -- what the shader clock of a SHORT, lone kernel is: a dependent chain of integer adds (one per 4+ cycles of a SIMD) timed
// with the constant 100 MHz counter (wall_clock64) and with clock64, launched (a) after the GPU sat idle for a few milliseconds -- the
// situation of a one-frame call -- and (b) right behind a long busy kernel.  hipcc -O3 --offload-arch=gfx950 tools/mb/clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void chain(int iters, long long* out, int* sink)
{
    int v = threadIdx.x;
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) v = v * 3 + i;      // dependent: v_mad_u32_u24 / mul+add chain
    }
    const long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; }
    if (v == 0x7fffffff) *sink = v;
}
__global__ void busy(int iters, int* sink)
{
    int v = threadIdx.x;
    for (int i = 0; i < iters; i++) v = v * 3 + i;
    if (v == 0x7fffffff) *sink = v;
}

int main()
{
    long long *d, h[2]; int* sink;
    CHECK(hipMalloc(&d, 16)); CHECK(hipMalloc(&sink, 4));
    const int iters = 40;          // 2 560 dependent ops: a few microseconds, like a phase of a one-workgroup kernel
    for (int rep = 0; rep < 3; rep++) {
        for (int mode = 0; mode < 3; mode++) {
            if (mode == 0) { CHECK(hipDeviceSynchronize()); usleep(5000); }
            if (mode == 1) { busy<<<1024, 256>>>(2000000, sink); }                 // ~ms of whole-chip work in front
            if (mode == 2) { CHECK(hipDeviceSynchronize()); usleep(200); }         // the gap between two calls of a frame loop
            chain<<<1, 64>>>(iters, d, sink);
            CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            const double ns = h[0] * 10.0, ops = iters * 64.0;
            printf("%-34s %6.2f us for %d dependent ops = %5.2f ns per op; clock64 %lld ticks = %.2f per ns\n",
                   mode == 0 ? "after 5 ms idle" : mode == 1 ? "behind a long busy kernel" : "after 0.2 ms idle", ns / 1e3, (int)ops, ns / ops, h[1], h[1] / ns);
        }
    }
    return 0;
}
