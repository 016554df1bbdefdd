// lds_fadd.hip -- does the LDS float atomic (ds_add_f32) round like v_add_f32?  Chains of positive adds (stamp-tap sized values) per
// address against the same chain in a register, bit for bit; then a few denormal cases (for information).  tools/mb.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(const float* __restrict__ v, int n, float* __restrict__ out_lds, float* __restrict__ out_reg)
{
    __shared__ float w[256];
    const int t = threadIdx.x, g = blockIdx.x * 256 + t;
    w[t] = 0.0f;
    __syncthreads();
    float acc = 0.0f;
    for (int i = 0; i < n; i++) {
        const float x = v[(size_t)i * gridDim.x * 256 + g];
        atomicAdd(&w[t], x);
        acc = acc + x;
    }
    __syncthreads();
    out_lds[g] = w[t]; out_reg[g] = acc;
}

int main()
{
    const int nb = 64, n = 4096, N = nb * 256;
    std::vector<float> h((size_t)n * N);
    srand(5);
    for (size_t i = 0; i < h.size(); i++) {
        const int kind = rand() % 8;
        const float u = (rand() & 0xffffff) / 16777216.0f;
        h[i] = kind == 0 ? u * 1e-9f : (kind == 1 ? u * 1e-4f : (kind == 2 ? u * 3.0f : u * 0.16f));     // taps of a sigma = 1 stamp: 1.2e-4 .. 0.16
    }
    // the last 256 chains: denormal operands / results
    for (int i = 0; i < n; i++) for (int t = 0; t < 256; t++) h[(size_t)i * N + (N - 256) + t] = (i < 8) ? 1e-41f * (t + 1) : 0.0f;
    float *d, *a, *b;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&a, N * 4)); CHECK(hipMalloc(&b, N * 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    k<<<nb, 256>>>(d, n, a, b);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ha(N), hb(N), hc(N);
    CHECK(hipMemcpy(ha.data(), a, N * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hb.data(), b, N * 4, hipMemcpyDeviceToHost));
    for (int g = 0; g < N; g++) { float acc = 0.0f; for (int i = 0; i < n; i++) acc = acc + h[(size_t)i * N + g]; hc[g] = acc; }
    int bad_lds = 0, bad_reg = 0, bad_den = 0;
    for (int g = 0; g < N - 256; g++) { bad_lds += memcmp(&ha[g], &hc[g], 4) != 0; bad_reg += memcmp(&hb[g], &hc[g], 4) != 0; }
    for (int g = N - 256; g < N; g++) bad_den += memcmp(&ha[g], &hc[g], 4) != 0;
    printf("chains of %d positive adds, %d addresses: ds_add_f32 vs host IEEE %d mismatches, v_add_f32 vs host %d mismatches\n", n, N - 256, bad_lds, bad_reg);
    printf("denormal chains (8 adds of k * 1e-41): ds_add_f32 vs host %d of 256 mismatches (lds %g, reg %g, host %g)\n", bad_den, ha[N - 1], hb[N - 1], hc[N - 1]);
    return 0;
}
