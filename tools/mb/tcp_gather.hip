// micro-benchmark: dwordx4 gathers from a 10 MB table: 64 distinct lines per wave instruction vs 16 (4 adjacent lanes share a 64-byte piece)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void gather(const float4* __restrict__ tab, const uint32_t* __restrict__ idx, int n_iter, int mode, float* out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    for (int it = 0; it < n_iter; it++) {
        uint32_t base;
        if (mode == 0) base = idx[(wave * n_iter + it) * 64 + lane];                       // every lane its own random 16-byte piece
        else if (mode == 1) base = (idx[(wave * n_iter + it) * 64 + (lane >> 2)] & ~3u) + (lane & 3);   // 4 adjacent lanes: one 64-byte block
        else base = (idx[(wave * n_iter + it) * 64 + (lane >> 3)] & ~7u) + (lane & 7);     // 8 adjacent lanes: one 128-byte line
        const float4 v = tab[base];
        acc += v.x + v.y + v.z + v.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    const size_t NT = (10u << 20) / 16;         // 10 MB of float4
    const int waves = 256 * 32, n_iter = 256;
    std::vector<uint32_t> h((size_t)waves * n_iter * 64);
    uint64_t s = 12345;
    for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint32_t)((s >> 33) % NT); }
    float4* tab; uint32_t* idx; float* out;
    hipMalloc(&tab, NT * 16); hipMemset(tab, 0, NT * 16);
    hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)waves * 64 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; mode++) {
        gather<<<waves / 4, 256>>>(tab, idx, n_iter, mode, out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) gather<<<waves / 4, 256>>>(tab, idx, n_iter, mode, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        const double instr = (double)waves * n_iter;
        printf("mode %d: %.3f ms, %.2f wave-loads/ns chip-wide, %.1f cycles per wave-load per CU (2.4 GHz), %.1f GB/s of requested bytes\n",
               mode, ms, instr / (ms * 1e6), ms * 1e-3 * 2.4e9 * 256 / instr, instr * 1024 / (ms * 1e6));
    }
    return 0;
}
