// slot_loop.hip -- per-entry cost of the slot gather's inner loop (tools/mb): one LDS row read + one add per entry, the row address
// taken from a byte of a dword held by another lane.  Variants isolate the pieces (see main).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(1024) void loop_kernel(const uint32_t* __restrict__ ent, int nlanes_iter, int reps, float* out, long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256 * 64; i += blockDim.x) lds[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    uint4 E = ((const uint4*)ent)[lane];
    const uint32_t lane4 = lane * 4u;
    uint32_t sel0 = 0x0c0c0400u, sel1 = 0x0c0c0500u, sel2 = 0x0c0c0600u, sel3 = 0x0c0c0700u;
    asm volatile("" : "+v"(sel0), "+v"(sel1), "+v"(sel2), "+v"(sel3));
    float acc = 0.f;
    const int lane_end = nlanes_iter;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        float r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
        uint32_t a0, a1, a2, a3;
        int sl, se;
        if (V == 0) {
#define SL_GROUP(EV, RA, RB, RC, RD, AA, AB, AC, AD) \
            "s_waitcnt lgkmcnt(8)\n" \
            "v_readlane_b32 %[se], %[" EV "], %[sl]\n" \
            "v_perm_b32 %[" RA "], %[se], %[l4], %[q0]\n" \
            "v_perm_b32 %[" RB "], %[se], %[l4], %[q1]\n" \
            "ds_read_b32 %[" RA "], %[" RA "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AA "]\n" \
            "v_perm_b32 %[" RC "], %[se], %[l4], %[q2]\n" \
            "ds_read_b32 %[" RB "], %[" RB "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AB "]\n" \
            "v_perm_b32 %[" RD "], %[se], %[l4], %[q3]\n" \
            "ds_read_b32 %[" RC "], %[" RC "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AC "]\n" \
            "ds_read_b32 %[" RD "], %[" RD "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AD "]\n"
#define BODY \
            asm volatile( \
                "v_mov_b32 %[r4], 0\n v_mov_b32 %[r5], 0\n v_mov_b32 %[r6], 0\n v_mov_b32 %[r7], 0\n" \
                "v_mov_b32 %[r8], 0\n v_mov_b32 %[r9], 0\n v_mov_b32 %[r10], 0\n v_mov_b32 %[r11], 0\n" \
                "v_mov_b32 %[r12], 0\n v_mov_b32 %[r13], 0\n v_mov_b32 %[r14], 0\n v_mov_b32 %[r15], 0\n" \
                "s_mov_b32 %[sl], 0\n" \
                "1:\n" \
                SL_GROUP("e0", "r0", "r1", "r2", "r3", "r4", "r5", "r6", "r7") \
                SL_GROUP("e1", "r4", "r5", "r6", "r7", "r8", "r9", "r10", "r11") \
                SL_GROUP("e2", "r8", "r9", "r10", "r11", "r12", "r13", "r14", "r15") \
                SL_GROUP("e3", "r12", "r13", "r14", "r15", "r0", "r1", "r2", "r3") \
                "s_add_u32 %[sl], %[sl], 1\n" \
                "s_cmp_lt_u32 %[sl], %[lend]\n" \
                "s_cbranch_scc1 1b\n" \
                "s_waitcnt lgkmcnt(0)\n" \
                "v_add_f32 %[acc], %[acc], %[r4]\n v_add_f32 %[acc], %[acc], %[r5]\n v_add_f32 %[acc], %[acc], %[r6]\n v_add_f32 %[acc], %[acc], %[r7]\n" \
                "v_add_f32 %[acc], %[acc], %[r8]\n v_add_f32 %[acc], %[acc], %[r9]\n v_add_f32 %[acc], %[acc], %[r10]\n v_add_f32 %[acc], %[acc], %[r11]\n" \
                "v_add_f32 %[acc], %[acc], %[r12]\n v_add_f32 %[acc], %[acc], %[r13]\n v_add_f32 %[acc], %[acc], %[r14]\n v_add_f32 %[acc], %[acc], %[r15]\n" \
                : [acc] "+v"(acc), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5), \
                  [r6] "=&v"(r6), [r7] "=&v"(r7), [r8] "=&v"(r8), [r9] "=&v"(r9), [r10] "=&v"(r10), [r11] "=&v"(r11), \
                  [r12] "=&v"(r12), [r13] "=&v"(r13), [r14] "=&v"(r14), [r15] "=&v"(r15), \
                  [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [sl] "=&s"(sl), [se] "=&s"(se) \
                : [e0] "v"(E.x), [e1] "v"(E.y), [e2] "v"(E.z), [e3] "v"(E.w), [lend] "s"(lane_end), [l4] "v"(lane4), \
                  [q0] "v"(sel0), [q1] "v"(sel1), [q2] "v"(sel2), [q3] "v"(sel3), [ldsp] "v"(lds) \
                : "scc", "memory");
            BODY
#undef SL_GROUP
        } else if (V == 1) {
            // no readlane / perm: fixed addresses (lane4 + constant rows)
#define SL_GROUP(EV, RA, RB, RC, RD, AA, AB, AC, AD) \
            "s_waitcnt lgkmcnt(8)\n" \
            "ds_read_b32 %[" RA "], %[l4] offset:256\n" \
            "v_add_f32 %[acc], %[acc], %[" AA "]\n" \
            "ds_read_b32 %[" RB "], %[l4] offset:512\n" \
            "v_add_f32 %[acc], %[acc], %[" AB "]\n" \
            "ds_read_b32 %[" RC "], %[l4] offset:1024\n" \
            "v_add_f32 %[acc], %[acc], %[" AC "]\n" \
            "ds_read_b32 %[" RD "], %[l4] offset:2048\n" \
            "v_add_f32 %[acc], %[acc], %[" AD "]\n"
            BODY
#undef SL_GROUP
        } else if (V == 2) {
            // separate address registers (a0..a3), data ring as before
#define SL_GROUP(EV, RA, RB, RC, RD, AA, AB, AC, AD) \
            "s_waitcnt lgkmcnt(8)\n" \
            "v_readlane_b32 %[se], %[" EV "], %[sl]\n" \
            "s_nop 1\n" \
            "v_perm_b32 %[a0], %[se], %[l4], %[q0]\n" \
            "v_perm_b32 %[a1], %[se], %[l4], %[q1]\n" \
            "v_perm_b32 %[a2], %[se], %[l4], %[q2]\n" \
            "v_perm_b32 %[a3], %[se], %[l4], %[q3]\n" \
            "ds_read_b32 %[" RA "], %[a0]\n" \
            "v_add_f32 %[acc], %[acc], %[" AA "]\n" \
            "ds_read_b32 %[" RB "], %[a1]\n" \
            "v_add_f32 %[acc], %[acc], %[" AB "]\n" \
            "ds_read_b32 %[" RC "], %[a2]\n" \
            "v_add_f32 %[acc], %[acc], %[" AC "]\n" \
            "ds_read_b32 %[" RD "], %[a3]\n" \
            "v_add_f32 %[acc], %[acc], %[" AD "]\n"
            BODY
#undef SL_GROUP
        } else if (V == 3) {
            // reads only (no adds inside the loop): LDS issue / latency alone
#define SL_GROUP(EV, RA, RB, RC, RD, AA, AB, AC, AD) \
            "s_waitcnt lgkmcnt(8)\n" \
            "ds_read_b32 %[" RA "], %[l4] offset:256\n" \
            "ds_read_b32 %[" RB "], %[l4] offset:512\n" \
            "ds_read_b32 %[" RC "], %[l4] offset:1024\n" \
            "ds_read_b32 %[" RD "], %[l4] offset:2048\n"
            BODY
#undef SL_GROUP
        } else if (V == 4) {
            // adds only
#define SL_GROUP(EV, RA, RB, RC, RD, AA, AB, AC, AD) \
            "v_add_f32 %[acc], %[acc], %[" AA "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AB "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AC "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AD "]\n"
            BODY
#undef SL_GROUP
        } else if (V == 5) {
            // readlane + perm + adds, no LDS
#define SL_GROUP(EV, RA, RB, RC, RD, AA, AB, AC, AD) \
            "v_readlane_b32 %[se], %[" EV "], %[sl]\n" \
            "v_perm_b32 %[a0], %[se], %[l4], %[q0]\n" \
            "v_perm_b32 %[a1], %[se], %[l4], %[q1]\n" \
            "v_add_f32 %[acc], %[acc], %[" AA "]\n" \
            "v_perm_b32 %[a2], %[se], %[l4], %[q2]\n" \
            "v_add_f32 %[acc], %[acc], %[" AB "]\n" \
            "v_perm_b32 %[a3], %[se], %[l4], %[q3]\n" \
            "v_add_f32 %[acc], %[acc], %[" AC "]\n" \
            "v_add_f32 %[acc], %[acc], %[" AD "]\n"
            BODY
#undef SL_GROUP
        } else if (V == 6) {
            // as V0 but wait for the group read FOUR groups ago is impossible (ring of 4): instead lgkmcnt(11): three groups in flight
#define SL_GROUP(EV, RA, RB, RC, RD, AA, AB, AC, AD) \
            "v_readlane_b32 %[se], %[" EV "], %[sl]\n" \
            "s_waitcnt lgkmcnt(11)\n" \
            "v_add_f32 %[acc], %[acc], %[" AA "]\n" \
            "v_perm_b32 %[" AA "], %[se], %[l4], %[q0]\n" \
            "ds_read_b32 %[" AA "], %[" AA "]\n" \
            "s_waitcnt lgkmcnt(11)\n" \
            "v_add_f32 %[acc], %[acc], %[" AB "]\n" \
            "v_perm_b32 %[" AB "], %[se], %[l4], %[q1]\n" \
            "ds_read_b32 %[" AB "], %[" AB "]\n" \
            "s_waitcnt lgkmcnt(11)\n" \
            "v_add_f32 %[acc], %[acc], %[" AC "]\n" \
            "v_perm_b32 %[" AC "], %[se], %[l4], %[q2]\n" \
            "ds_read_b32 %[" AC "], %[" AC "]\n" \
            "s_waitcnt lgkmcnt(11)\n" \
            "v_add_f32 %[acc], %[acc], %[" AD "]\n" \
            "v_perm_b32 %[" AD "], %[se], %[l4], %[q3]\n" \
            "ds_read_b32 %[" AD "], %[" AD "]\n"
            BODY
#undef SL_GROUP
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* what, int nwg, int nthr, int reps, const uint32_t* d_ent, float* d_out, long long* d_cyc)
{
    CHECK(hipFuncSetAttribute((const void*)loop_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    loop_kernel<V><<<nwg, nthr, 64 * 1024>>>(d_ent, 64, 4, d_out, d_cyc);
    CHECK(hipEventRecord(a));
    loop_kernel<V><<<nwg, nthr, 64 * 1024>>>(d_ent, 64, reps, d_out, d_cyc);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    long long cyc; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    const double entries_per_wave = 1024.0 * reps;
    const double total = entries_per_wave * nwg * (nthr / 64);
    printf("%-46s wg=%4d thr=%4d: %7.3f ms  %7.2f ns/entry/wave  clock64 ticks/entry (wave 0) %7.2f  chip %8.2f Gentries/s\n", what, nwg, nthr, ms,
           ms * 1e6 / entries_per_wave, (double)cyc / entries_per_wave, total / ms * 1e-6);
}

int main()
{
    std::vector<uint32_t> h(256);
    srand(1);
    for (auto& v : h) v = ((rand() % 200) | ((rand() % 200) << 8) | ((rand() % 200) << 16) | ((uint32_t)(rand() % 200) << 24));
    uint32_t* d_ent; float* d_out; long long* d_cyc;
    CHECK(hipMalloc(&d_ent, 1024)); CHECK(hipMalloc(&d_out, 4 << 20)); CHECK(hipMalloc(&d_cyc, 8 << 12));
    CHECK(hipMemcpy(d_ent, h.data(), 1024, hipMemcpyHostToDevice));
    const int reps = 200;
    for (int cfg = 0; cfg < 3; cfg++) {
        const int nwg = cfg == 0 ? 1 : 512, nthr = cfg == 2 ? 1024 : (cfg == 1 ? 256 : 64);
        run<0>("V0 readlane+perm+read(dst=addr)+add", nwg, nthr, reps, d_ent, d_out, d_cyc);
        run<2>("V2 separate address registers, nop after readlane", nwg, nthr, reps, d_ent, d_out, d_cyc);
        run<1>("V1 fixed addresses: read+add", nwg, nthr, reps, d_ent, d_out, d_cyc);
        run<3>("V3 reads only", nwg, nthr, reps, d_ent, d_out, d_cyc);
        run<4>("V4 adds only", nwg, nthr, reps, d_ent, d_out, d_cyc);
        run<5>("V5 readlane+perm+add, no LDS", nwg, nthr, reps, d_ent, d_out, d_cyc);
    }
    return 0;
}
