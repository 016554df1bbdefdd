/* check_libm.c -- ORACLE tooling: exhaustive comparison of orc_expf / orc_sinf / orc_cosf / orc_tanf / orc_atanf (and of
 * orc_atan2f on 400 M generated pairs) with the host libm.
 *   gcc -O2 -ffp-contract=off -fopenmp -o check_libm check_libm.c orc_math.c -lm && ./check_libm     (about a minute on 8 cores) */
#include "eorb_oracle.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float fromb(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(void)
{
    unsigned long long n = 0, bad = 0;
    for (uint32_t u = bits(-0x1p-30f); u <= bits(-104.0f); u++) {
        float x = fromb(u);
        n++;
        if (bits(orc_expf(x)) != bits(expf(x))) { if (bad < 5) printf("expf differs at x=%a: oracle %a libm %a\n", x, orc_expf(x), expf(x)); bad++; }
    }
    printf("expf: %llu inputs in [-104, -2^-30], %llu differ from the host libm\n", n, bad);
    unsigned long long bs = 0, bc = 0; n = 0;
    for (uint32_t u = bits(0x1p-20f); u <= bits(6.5f); u++) {
        float x = fromb(u);
        n++;
        if (bits(orc_sinf(x)) != bits(sinf(x))) bs++;
        if (bits(orc_cosf(x)) != bits(cosf(x))) bc++;
        if (bits(orc_sinf(-x)) != bits(sinf(-x))) bs++;
        if (bits(orc_cosf(-x)) != bits(cosf(-x))) bc++;
    }
    printf("sinf/cosf: %llu inputs in [2^-20, 6.5] and their negatives, %llu / %llu differ from the host libm\n", 2 * n, bs, bc);
    unsigned long long bt = 0, ba = 0, b2 = 0, nt = 0, na = 0, n2 = 0;
    const uint32_t thi = bits(2.35f);
#pragma omp parallel for reduction(+:bt, nt) schedule(static)
    for (uint32_t u = 0; u <= thi; u++) {
        const float x = fromb(u);
        nt += 2;
        if (bits(orc_tanf(x)) != bits(tanf(x))) bt++;
        if (bits(orc_tanf(-x)) != bits(tanf(-x))) bt++;
    }
    printf("tanf: %llu inputs in [-2.35, 2.35], %llu differ from the host libm\n", nt, bt);
#pragma omp parallel for reduction(+:ba, na) schedule(static)
    for (uint32_t u = 0; u < 0x7f800000u; u++) {
        const float x = fromb(u);
        na++;
        if (bits(orc_atanf(x)) != bits(atanf(x))) ba++;
    }
    printf("atanf: %llu positive inputs, %llu differ from the host libm\n", na, ba);
#pragma omp parallel for reduction(+:b2, n2) schedule(static)
    for (long long i = 0; i < 400000000LL; i++) {
        float y, x; orc_atan2_pair((uint64_t)i, &y, &x);
        if (y != y || x != x) continue;
        n2++;
        if (bits(orc_atan2f(y, x)) != bits(atan2f(y, x))) b2++;
    }
    printf("atan2f: %llu generated pairs, %llu differ from the host libm\n", n2, b2);
    return 0;
}
