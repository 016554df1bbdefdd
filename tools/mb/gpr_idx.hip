// gpr_idx.hip -- cost of "row in a register, picked by s_set_gpr_idx_idx": per entry one SALU (index) + one VALU (add), tools/mb.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ ent, int nd, int reps, float* out, long long* cyc, float* chk)
{
    const int lane = threadIdx.x;
    float acc = 0.f;
    const uint32_t* e = ent;
    long long t0 = clock64();
    // rows v64..v95 = (row + 1) * 0.5 + lane * 1e-3
    asm volatile(
        "v_cvt_f32_u32 v96, %[lane]\n v_mul_f32 v96, 0x3a83126f, v96\n"
        "v_add_f32 v64, 0.5, v96\n v_add_f32 v65, 1.0, v96\n v_add_f32 v66, 0.5, v65\n v_add_f32 v67, 0.5, v66\n"
        "v_add_f32 v68, 0.5, v67\n v_add_f32 v69, 0.5, v68\n v_add_f32 v70, 0.5, v69\n v_add_f32 v71, 0.5, v70\n"
        "s_mov_b32 s40, 0\n"
        "s_set_gpr_idx_on s40, gpr_idx(SRC0)\n"
        "1:\n"
        "s_load_dwordx8 s[44:51], %[e], 0x0\n"
        "s_waitcnt lgkmcnt(0)\n"
        ".macro ONE sreg\n"
        "s_set_gpr_idx_idx \\sreg\n v_add_f32 %[acc], v64, %[acc]\n"
        "s_lshr_b32 s41, \\sreg, 8\n s_set_gpr_idx_idx s41\n v_add_f32 %[acc], v64, %[acc]\n"
        "s_lshr_b32 s41, \\sreg, 16\n s_set_gpr_idx_idx s41\n v_add_f32 %[acc], v64, %[acc]\n"
        "s_lshr_b32 s41, \\sreg, 24\n s_set_gpr_idx_idx s41\n v_add_f32 %[acc], v64, %[acc]\n"
        ".endm\n"
        "ONE s44\n ONE s45\n ONE s46\n ONE s47\n ONE s48\n ONE s49\n ONE s50\n ONE s51\n"
        ".purgem ONE\n"
        "s_add_u32 s40, s40, 1\n"
        "s_cmp_lt_u32 s40, %[reps]\n"
        "s_cbranch_scc1 1b\n"
        "s_set_gpr_idx_off\n"
        : [acc] "+v"(acc)
        : [e] "s"(e), [reps] "s"(reps), [lane] "v"(lane)
        : "s40", "s41", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v96", "m0", "scc", "memory");
    long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    (void)nd; (void)chk;
}

int main()
{
    uint32_t h[8];
    srand(3);
    for (int i = 0; i < 8; i++) h[i] = (rand() % 8) | ((rand() % 8) << 8) | ((rand() % 8) << 16) | ((uint32_t)(rand() % 8) << 24);
    uint32_t* d; float* out; long long* cyc;
    CHECK(hipMalloc(&d, 64)); CHECK(hipMalloc(&out, 4 << 20)); CHECK(hipMalloc(&cyc, 8 << 12));
    CHECK(hipMemcpy(d, h, 32, hipMemcpyHostToDevice));
    const int reps = 20000;
    for (int nwg : {1, 256, 1024}) {
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        k<<<nwg, 64>>>(d, 8, 100, out, cyc, nullptr);
        CHECK(hipEventRecord(a));
        k<<<nwg, 64>>>(d, 8, reps, out, cyc, nullptr);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        float o[64]; CHECK(hipMemcpy(o, out, 256, hipMemcpyDeviceToHost));
        // reference on the host
        float ref = 0.f, ref5 = 0.f;
        for (int r = 0; r < reps; r++) for (int i = 0; i < 8; i++) for (int b8 = 0; b8 < 4; b8++) {
            const int idx = (h[i] >> (8 * b8)) & 0xff;
            ref = ref + ((idx + 1) * 0.5f + 0 * 1e-3f); ref5 = ref5 + ((idx + 1) * 0.5f + 5 * 0.001f);
        }
        printf("wg=%4d: %.3f ms, %.2f clock64 ticks per entry (wave 0), %.2f ns per entry per wave; lane0 %.3f (host %.3f)\n", nwg, ms,
               (double)c / (32.0 * reps), ms * 1e6 / (32.0 * reps), o[0], ref);
    }
    return 0;
}
