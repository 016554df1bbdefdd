/* check_libm.c -- ORACLE tooling: exhaustive comparison of orc_expf / orc_sinf / orc_cosf with the host libm.
 *   gcc -O2 -ffp-contract=off -o check_libm check_libm.c orc_math.c -lm && ./check_libm     (about 10 s) */
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
    }
    printf("sinf/cosf: %llu inputs in [2^-20, 6.5], %llu / %llu differ from the host libm\n", n, bs, bc);
    return 0;
}
