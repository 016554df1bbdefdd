// lds_atomic_order.hip -- does a wave's ds_add_rtn_u32 hand out its results in LANE ORDER among lanes that hit the same word?
// (needed by a stable scatter that takes ranks from LDS atomics).  Random collision patterns, 32-bit counters and 16-bit halves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k(const uint32_t* addr, int ntrial, int naddr_mask_shift, unsigned long long* bad, int packed, int partial)
{
    __shared__ uint32_t cnt[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long nbad = 0;
    for (int t = 0; t < ntrial; t++) {
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) cnt[i] = 0;
        __syncthreads();
        uint32_t a = addr[(blockIdx.x * ntrial + t) * blockDim.x + threadIdx.x];
        const bool act = !partial || ((a >> 20) & 3) != 0;       // some lanes inactive
        a = (a & 0xffff) >> naddr_mask_shift;                     // address range 2^(16-shift)
        a = a % 120 + wave * 128;                                 // every wave its own words
        uint32_t r = 0xffffffffu;
        if (act) {
            if (packed) { const uint32_t o = atomicAdd(&cnt[a >> 1], 1u << (16 * (a & 1))); r = (o >> (16 * (a & 1))) & 0xffff; }
            else r = atomicAdd(&cnt[a], 1u);
        }
        // lanes with the same address: ranks must rise with the lane index
        for (int j = 0; j < 64; j++) {
            const uint32_t aj = __shfl(a, j, 64), rj = __shfl(r, j, 64);
            if (act && rj != 0xffffffffu && j < lane && aj == a && !(rj < r)) nbad++;
        }
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main()
{
    const int nwg = 512, nthr = 256, ntrial = 64;
    const size_t n = (size_t)nwg * ntrial * nthr;
    uint32_t* h = (uint32_t*)malloc(n * 4);
    srand(7);
    for (size_t i = 0; i < n; i++) h[i] = ((uint32_t)rand() << 8) ^ (uint32_t)rand();
    uint32_t* d; unsigned long long* bad;
    CHECK(hipMalloc(&d, n * 4)); CHECK(hipMalloc(&bad, 8));
    CHECK(hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice));
    for (int packed = 0; packed < 2; packed++)
        for (int partial = 0; partial < 2; partial++)
            for (int shift = 9; shift <= 15; shift += 2) {
                CHECK(hipMemset(bad, 0, 8));
                k<<<nwg, nthr>>>(d, ntrial, shift, bad, packed, partial);
                unsigned long long b; CHECK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
                printf("packed=%d partial=%d distinct addresses per wave <= %d: order violations %llu\n", packed, partial, 1 << (16 - shift), b);
            }
    return 0;
}
