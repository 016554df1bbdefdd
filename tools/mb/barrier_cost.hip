// barrier_cost.hip -- what a workgroup barrier and HIP's __syncthreads_or cost with 256 / 1024 threads (cycles per iteration of a loop
// that does nothing else), and a hand-made "any thread changed?" (one LDS flag, two barriers).  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k(long long* out, int iters, int* sink)
{
    __shared__ int flag[2];
    int v = threadIdx.x;
    if (threadIdx.x < 2) flag[threadIdx.x] = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { __syncthreads(); v += i; }
        else if (MODE == 1) { v += __syncthreads_or((v & 1023) == 5000 + i); }
        else if (MODE == 2) {                    // flag of this iteration's parity; the other one is cleared for the next
            if ((v & 1023) == 5000 + i) flag[i & 1] = 1;
            __syncthreads();
            v += flag[i & 1];
            if (threadIdx.x == 0) flag[(i + 1) & 1] = 0;
        } else { __syncthreads(); __syncthreads(); v += __syncthreads_or((v & 1023) == 5000 + i); }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == -12345) *sink = v;
}

int main()
{
    long long* d; int* sink; CHECK(hipMalloc(&d, 64)); CHECK(hipMalloc(&sink, 4));
    const int iters = 200;
    for (int threads : {256, 1024}) {
        long long h;
#define RUN(M, name) k<M><<<1, threads>>>(d, iters, sink); CHECK(hipDeviceSynchronize()); k<M><<<1, threads>>>(d, iters, sink); CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); \
        printf("%4d threads  %-44s %7.0f cycles per iteration\n", threads, name, (double)h / iters);
        RUN(0, "__syncthreads()")
        RUN(1, "__syncthreads_or()")
        RUN(2, "LDS flag + one __syncthreads()")
        RUN(3, "2 x __syncthreads() + __syncthreads_or()")
    }
    return 0;
}
