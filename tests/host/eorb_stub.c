/* A stand-in for the five entry points of libeorb_fe.so that eorb_host::Context / ContextPool call, so that the pool's own locking
 * can run under ThreadSanitizer on a machine without a GPU (tests/test_oracle_hygiene.py).  Not a CPU path of the product: it
 * computes nothing. */
#include <stdlib.h>
#include <string.h>
#include "../../include/eorb_fe.h"

struct eorb_ctx { int uploads; int last_lw; float last_first; int fail_next; };
static int g_fail_once = 0;
int g_inconsistent = 0;

void eorb_stub_fail_once(void) { __atomic_store_n(&g_fail_once, 1, __ATOMIC_SEQ_CST); }
int eorb_stub_inconsistent(void) { return __atomic_load_n(&g_inconsistent, __ATOMIC_SEQ_CST); }

int eorb_create(int device, void* stream, eorb_ctx** out) { (void)device; (void)stream; *out = (eorb_ctx*)calloc(1, sizeof(eorb_ctx)); return *out ? EORB_OK : EORB_E_HIP; }
void eorb_destroy(eorb_ctx* c) { free(c); }
const char* eorb_last_error(eorb_ctx* c) { (void)c; return "stub failure"; }
int eorb_set_undistort_maps(eorb_ctx* c, const float* mx, const float* my, int LW, int LH, int check)
{
    (void)my; (void)LH; (void)check;
    int one = 1;
    if (__atomic_compare_exchange_n(&g_fail_once, &one, 0, 0, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return EORB_E_HIP;
    /* the snapshot handed over must be a consistent pair: the driver writes mapX[0] = mapX[LW*LH-1] = LW */
    if (mx[0] != (float)LW || mx[(size_t)LW * LH - 1] != (float)LW) __atomic_store_n(&g_inconsistent, 1, __ATOMIC_SEQ_CST);
    c->uploads++; c->last_lw = LW; c->last_first = mx[0];
    return EORB_OK;
}
int eorb_bow_set_vocabulary(eorb_ctx* c, int nnodes, int L, const int32_t* child_off, const int32_t* child_ids, const uint8_t* node_desc,
                            const int32_t* word_id, const double* weight)
{
    (void)c; (void)nnodes; (void)L; (void)child_off; (void)child_ids; (void)node_desc; (void)word_id; (void)weight;
    return EORB_OK;
}
