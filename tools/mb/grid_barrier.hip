// grid_barrier.hip -- what a grid-wide barrier inside ONE launch costs (an atomic counter in global memory, every workgroup's thread 0 adds
// and polls, bounded spin), against the ~4 us a dependent launch costs in a chain: would the pyramid's levels be cheaper as phases of one
// kernel?  G workgroups (all resident: G <= number of CUs), K barriers.  hipcc -O3 --offload-arch=gfx950 tools/mb/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned int* ctr, unsigned int target)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(ctr, 1u);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) { if (++spins > (1 << 22)) { ok = false; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    return ok;
}
__global__ void k_barriers(unsigned int* ctr, int K, long long* out, unsigned char* buf, int n)
{
    const long long t0 = wall_clock64();
    for (int k = 0; k < K; k++) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = (unsigned char)(buf[i] + k);      // a level's worth of work
        if (!grid_barrier(ctr, (unsigned)(k + 1) * gridDim.x)) break;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = wall_clock64() - t0;
}
__global__ void k_level(unsigned char* buf, int n, int k)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = (unsigned char)(buf[i] + k);
}
int main()
{
    unsigned int* ctr; long long *d, h; unsigned char* buf; const int n = 240 * 180;
    CHECK(hipMalloc(&ctr, 4)); CHECK(hipMalloc(&d, 8)); CHECK(hipMalloc(&buf, n)); CHECK(hipMemset(buf, 0, n));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int G : {32, 64, 128, 200}) {
        for (int K : {1, 4, 8}) {
            float best = 1e9f; long long bt = 0;
            for (int rep = 0; rep < 20; rep++) {
                CHECK(hipMemset(ctr, 0, 4));
                CHECK(hipEventRecord(e0));
                k_barriers<<<G, 256>>>(ctr, K, d, buf, n);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
                if (ms < best) { best = ms; bt = h; }
            }
            float chain = 1e9f;
            for (int rep = 0; rep < 20; rep++) {
                CHECK(hipEventRecord(e0));
                for (int k = 0; k < K; k++) k_level<<<G, 256>>>(buf, n, k);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < chain) chain = ms;
            }
            printf("%3d workgroups, %d phases: one launch with grid barriers %6.1f us (in-kernel %5.1f us), %d dependent launches %6.1f us\n", G, K, best * 1e3, bt * 0.01, K, chain * 1e3);
        }
    }
    return 0;
}
