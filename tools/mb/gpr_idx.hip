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

// 16-bit entries (slot | 0x1000: M0[12] = the SRC0 enable of the index mode): ONE scalar instruction per entry writes M0 whole
__global__ __launch_bounds__(64) void k16(const uint32_t* __restrict__ ent, int nd, int reps, float* out, long long* cyc, float* chk)
{
    const int lane = threadIdx.x;
    float acc = 0.f;
    const uint32_t* e = ent;
    long long t0 = clock64();
    asm volatile(
        "v_cvt_f32_u32 v96, %[lane]\n v_mul_f32 v96, 0x3a83126f, v96\n"
        "v_add_f32 v64, 0.5, v96\n v_add_f32 v65, 1.0, v96\n v_add_f32 v66, 0.5, v65\n v_add_f32 v67, 0.5, v66\n"
        "v_add_f32 v68, 0.5, v67\n v_add_f32 v69, 0.5, v68\n v_add_f32 v70, 0.5, v69\n v_add_f32 v71, 0.5, v70\n"
        "s_mov_b32 s40, 0\n"
        "s_set_gpr_idx_on s40, gpr_idx(SRC0)\n"
        "1:\n"
        "s_load_dwordx16 s[44:59], %[e], 0x0\n"
        "s_waitcnt lgkmcnt(0)\n"
        ".macro TWO sreg\n"
        "s_mov_b32 m0, \\sreg\n v_add_f32 %[acc], v64, %[acc]\n"
        "s_lshr_b32 m0, \\sreg, 16\n v_add_f32 %[acc], v64, %[acc]\n"
        ".endm\n"
        "TWO s44\n TWO s45\n TWO s46\n TWO s47\n TWO s48\n TWO s49\n TWO s50\n TWO s51\n"
        "TWO s52\n TWO s53\n TWO s54\n TWO s55\n TWO s56\n TWO s57\n TWO s58\n TWO s59\n"
        ".purgem TWO\n"
        "s_add_u32 s40, s40, 1\n"
        "s_cmp_lt_u32 s40, %[reps]\n"
        "s_cbranch_scc1 1b\n"
        "s_set_gpr_idx_off\n"
        : [acc] "+v"(acc)
        : [e] "s"(e), [reps] "s"(reps), [lane] "v"(lane)
        : "s40", "s41", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59",
          "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v96", "m0", "scc", "memory");
    long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    (void)nd; (void)chk;
}

// entries in VGPRs (as after a vector load), handed to the scalar side by v_readlane_b32 one lane ahead; index mode on source 1
__global__ __launch_bounds__(64) void kv_inter(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, int reps, float* out, long long* cyc)
{
    const int lane = threadIdx.x;
    float acc = 0.f;
    long long t0 = clock64();
    asm volatile(
        "v_cvt_f32_u32 v96, %[lane]\n"
        "v_mul_f32 v96, 0x3a83126f, v96\n"
        "v_add_f32 v64, 0.5, v96\n"
        "v_add_f32 v65, 1.0, v96\n"
        "v_add_f32 v66, 0.5, v65\n"
        "v_add_f32 v67, 0.5, v66\n"
        "v_add_f32 v68, 0.5, v67\n"
        "v_add_f32 v69, 0.5, v68\n"
        "v_add_f32 v70, 0.5, v69\n"
        "v_add_f32 v71, 0.5, v70\n"
        "v_mov_b32 v100, %[e0]\n"
        "v_mov_b32 v101, %[e1]\n"
        "v_mov_b32 v102, %[e2]\n"
        "v_mov_b32 v103, %[e3]\n"
        "s_mov_b32 s40, 0\n"
        "v_readlane_b32 s44, v100, 0\n"
        "v_readlane_b32 s45, v101, 0\n"
        "v_readlane_b32 s46, v102, 0\n"
        "v_readlane_b32 s47, v103, 0\n"
        "s_mov_b32 s41, 0\n"
        "s_set_gpr_idx_on s41, gpr_idx(SRC1)\n"
        "1:\n"
        "s_add_u32 s42, s40, 1\n"
        "v_readlane_b32 s48, v100, s42\n"
        "s_mov_b32 m0, s44\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s44, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "v_readlane_b32 s49, v101, s42\n"
        "s_mov_b32 m0, s45\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s45, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "v_readlane_b32 s50, v102, s42\n"
        "s_mov_b32 m0, s46\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s46, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "v_readlane_b32 s51, v103, s42\n"
        "s_mov_b32 m0, s47\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s47, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_add_u32 s40, s40, 2\n"
        "v_readlane_b32 s44, v100, s40\n"
        "s_mov_b32 m0, s48\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s48, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "v_readlane_b32 s45, v101, s40\n"
        "s_mov_b32 m0, s49\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s49, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "v_readlane_b32 s46, v102, s40\n"
        "s_mov_b32 m0, s50\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s50, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "v_readlane_b32 s47, v103, s40\n"
        "s_mov_b32 m0, s51\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s51, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_cmp_lt_u32 s40, %[reps]\n"
        "s_cbranch_scc1 1b\n"
        "s_set_gpr_idx_off\n"
        : [acc] "+v"(acc)
        : [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [e3] "s"(e3), [reps] "s"(reps), [lane] "v"(lane)
        : "s40", "s41", "s42", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v96",
          "v100", "v101", "v102", "v103", "m0", "scc", "memory");
    long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

// entries in VGPRs (as after a vector load), handed to the scalar side by v_readlane_b32 one lane ahead; index mode on source 1
__global__ __launch_bounds__(64) void kv_batch(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, int reps, float* out, long long* cyc)
{
    const int lane = threadIdx.x;
    float acc = 0.f;
    long long t0 = clock64();
    asm volatile(
        "v_cvt_f32_u32 v96, %[lane]\n"
        "v_mul_f32 v96, 0x3a83126f, v96\n"
        "v_add_f32 v64, 0.5, v96\n"
        "v_add_f32 v65, 1.0, v96\n"
        "v_add_f32 v66, 0.5, v65\n"
        "v_add_f32 v67, 0.5, v66\n"
        "v_add_f32 v68, 0.5, v67\n"
        "v_add_f32 v69, 0.5, v68\n"
        "v_add_f32 v70, 0.5, v69\n"
        "v_add_f32 v71, 0.5, v70\n"
        "v_mov_b32 v100, %[e0]\n"
        "v_mov_b32 v101, %[e1]\n"
        "v_mov_b32 v102, %[e2]\n"
        "v_mov_b32 v103, %[e3]\n"
        "s_mov_b32 s40, 0\n"
        "v_readlane_b32 s44, v100, 0\n"
        "v_readlane_b32 s45, v101, 0\n"
        "v_readlane_b32 s46, v102, 0\n"
        "v_readlane_b32 s47, v103, 0\n"
        "s_mov_b32 s41, 0\n"
        "s_set_gpr_idx_on s41, gpr_idx(SRC1)\n"
        "1:\n"
        "s_add_u32 s42, s40, 1\n"
        "v_readlane_b32 s48, v100, s42\n"
        "v_readlane_b32 s49, v101, s42\n"
        "v_readlane_b32 s50, v102, s42\n"
        "v_readlane_b32 s51, v103, s42\n"
        "s_mov_b32 m0, s44\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s44, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_mov_b32 m0, s45\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s45, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_mov_b32 m0, s46\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s46, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_mov_b32 m0, s47\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s47, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_add_u32 s40, s40, 2\n"
        "v_readlane_b32 s44, v100, s40\n"
        "v_readlane_b32 s45, v101, s40\n"
        "v_readlane_b32 s46, v102, s40\n"
        "v_readlane_b32 s47, v103, s40\n"
        "s_mov_b32 m0, s48\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s48, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_mov_b32 m0, s49\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s49, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_mov_b32 m0, s50\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s50, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_mov_b32 m0, s51\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_lshr_b32 m0, s51, 16\n"
        "v_add_f32 %[acc], %[acc], v64\n"
        "s_cmp_lt_u32 s40, %[reps]\n"
        "s_cbranch_scc1 1b\n"
        "s_set_gpr_idx_off\n"
        : [acc] "+v"(acc)
        : [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [e3] "s"(e3), [reps] "s"(reps), [lane] "v"(lane)
        : "s40", "s41", "s42", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v96",
          "v100", "v101", "v102", "v103", "m0", "scc", "memory");
    long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
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
    for (int nwg : {1, 256, 512, 1024, 2048}) {
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
    {
        uint32_t h16[16];
        for (int i = 0; i < 16; i++) h16[i] = (0x1000u | (rand() % 8)) | ((0x1000u | (uint32_t)(rand() % 8)) << 16);
        uint32_t* d16; CHECK(hipMalloc(&d16, 64)); CHECK(hipMemcpy(d16, h16, 64, hipMemcpyHostToDevice));
        for (int nwg : {1, 256, 512, 1024, 2048}) {
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            k16<<<nwg, 64>>>(d16, 16, 100, out, cyc, nullptr);
            CHECK(hipEventRecord(a));
            k16<<<nwg, 64>>>(d16, 16, reps, out, cyc, nullptr);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            float o[64]; CHECK(hipMemcpy(o, out, 256, hipMemcpyDeviceToHost));
            float ref = 0.f;
            for (int r = 0; r < reps; r++) for (int i = 0; i < 16; i++) for (int b2 = 0; b2 < 2; b2++) {
                const int idx = (h16[i] >> (16 * b2)) & 0xff;
                ref = ref + ((idx + 1) * 0.5f + 0 * 1e-3f);
            }
            printf("16-bit entries wg=%4d: %.3f ms, %.2f clock64 ticks per entry (wave 0), %.2f ns per entry per wave; lane0 %.3f (host %.3f) %s\n", nwg, ms,
                   (double)c / (32.0 * reps), ms * 1e6 / (32.0 * reps), o[0], ref, o[0] == ref ? "OK" : "MISMATCH");
        }
    }
    for (int variant = 0; variant < 2; variant++) {
        const uint32_t e[4] = {0x20012003u, 0x20052000u, 0x20072002u, 0x20042006u};
        for (int nwg : {1, 1024, 2048}) {
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            const int lanes = 2 * 20000;      // "lanes" walked (8 entries each)
            if (variant) kv_batch<<<nwg, 64>>>(e[0], e[1], e[2], e[3], 100, out, cyc); else kv_inter<<<nwg, 64>>>(e[0], e[1], e[2], e[3], 100, out, cyc);
            CHECK(hipEventRecord(a));
            if (variant) kv_batch<<<nwg, 64>>>(e[0], e[1], e[2], e[3], lanes, out, cyc); else kv_inter<<<nwg, 64>>>(e[0], e[1], e[2], e[3], lanes, out, cyc);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            float o[64]; CHECK(hipMemcpy(o, out, 256, hipMemcpyDeviceToHost));
            float ref = 0.f;
            for (int l = 0; l < lanes; l++) for (int i = 0; i < 4; i++) for (int b2 = 0; b2 < 2; b2++) ref = ref + ((((e[i] >> (16 * b2)) & 0xff) + 1) * 0.5f);
            printf("%s wg=%4d: %.3f ms, %.2f clock64 ticks per entry (wave 0), %.2f ns per entry per wave; lane0 %.3f (host %.3f) %s\n", variant ? "readlane batched    " : "readlane interleaved",
                   nwg, ms, (double)c / (8.0 * lanes), ms * 1e6 / (8.0 * lanes), o[0], ref, o[0] == ref ? "OK" : "MISMATCH");
        }
    }
    return 0;
}
