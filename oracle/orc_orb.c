/*
 * orc_orb.c -- ORACLE (test infrastructure only): ORB extractor.
 * Restates src/ORBextractor.cc of the reference plus the OpenCV 3.4.1 primitives it calls
 * (cv::resize INTER_LINEAR 8u, copyMakeBorder REFLECT_101, cv::FAST TYPE_9_16 + NMS,
 * GaussianBlur 5x5 sigma 2 8u, fastAtan2, cvRound).  OpenCV is not vendored by the reference and
 * is not in this image: those primitives are restated from their published 3.4.x algorithms and
 * are "parity unpinned" (oracle/README.md lists every choice).
 */
#include "eorb_oracle.h"
#include "orc_pattern.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_SIZE 31
#define HALF_PATCH_SIZE 15
#define W_DENOM 30.0f
#define MAX_LEVELS 16

struct orc_orb {
    orc_orb_params p;
    int   edge;                         /* per-instance EDGE_THRESHOLD (SURVEY App.B H3) */
    float sf[MAX_LEVELS], inv_sf[MAX_LEVELS], sigma2[MAX_LEVELS], inv_sigma2[MAX_LEVELS];
    int   nfeat_level[MAX_LEVELS];
    int   umax[HALF_PATCH_SIZE + 1];
    /* last call state */
    int      lw[MAX_LEVELS], lh[MAX_LEVELS];
    uint8_t* buf[MAX_LEVELS];           /* bordered buffers (lw+2e)x(lh+2e) */
    uint8_t* blur[MAX_LEVELS];          /* lw x lh or NULL */
    orc_keypoint* cand[MAX_LEVELS]; int ncand[MAX_LEVELS];
    orc_keypoint* kps[MAX_LEVELS];  int nkps[MAX_LEVELS];
};

static inline int cv_floor(double v) { return (int)floor(v); }
static inline int cv_ceil(double v) { return (int)ceil(v); }

/* ---- ctor: src/ORBextractor.cc:420-489 ------------------------------------------------------ */
orc_orb* orc_orb_create(const orc_orb_params* p)
{
    if (!p || p->nlevels < 1 || p->nlevels > MAX_LEVELS) return NULL;
    orc_orb* e = (orc_orb*)calloc(1, sizeof(orc_orb));
    e->p = *p;
    const int nlevels = p->nlevels;
    const double scaleFactor = (double)p->scaleFactor;        /* float param -> double member */
    e->sf[0] = 1.0f; e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->sf[i] = (float)((double)e->sf[i - 1] * scaleFactor);
        e->sigma2[i] = e->sf[i] * e->sf[i];
    }
    for (int i = 0; i < nlevels; i++) {
        e->inv_sf[i] = 1.0f / e->sf[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    float factor = (float)(1.0 / scaleFactor);
    float nDesired = (float)p->nfeatures * (1 - factor) /
                     (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        e->nfeat_level[l] = orc_cvround((double)nDesired);
        sum += e->nfeat_level[l];
        nDesired *= factor;
    }
    e->nfeat_level[nlevels - 1] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;

    /* umax :463-478 */
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = orc_cvround(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    /* EDGE_THRESHOLD :481-488 (global default 19, header DEF_IMAGE_WIDTH 752) */
    if (p->edgeTh < 0) {
        float newEdge = 19 * ((float)p->imWidth / (float)752);
        e->edge = (int)newEdge;
        e->edge += (e->edge % 2 - 1);
    } else {
        e->edge = p->edgeTh;
    }
    return e;
}

static void free_state(orc_orb* e)
{
    for (int l = 0; l < MAX_LEVELS; l++) {
        free(e->buf[l]); e->buf[l] = NULL;
        free(e->blur[l]); e->blur[l] = NULL;
        free(e->cand[l]); e->cand[l] = NULL; e->ncand[l] = 0;
        free(e->kps[l]); e->kps[l] = NULL; e->nkps[l] = 0;
    }
}
void orc_orb_destroy(orc_orb* e) { if (e) { free_state(e); free(e); } }

int orc_orb_edge_threshold(const orc_orb* e) { return e->edge; }
const float* orc_orb_scale_factors(const orc_orb* e) { return e->sf; }
const float* orc_orb_inv_scale_factors(const orc_orb* e) { return e->inv_sf; }
const int* orc_orb_features_per_level(const orc_orb* e) { return e->nfeat_level; }
const int* orc_orb_umax(const orc_orb* e) { return e->umax; }
int orc_orb_max_keypoints(const orc_orb* e) { return e->p.nfeatures + 8 * e->p.nlevels; }

/* ---- OpenCV: borderInterpolate(REFLECT_101) -------------------------------------------------- */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* ---- OpenCV 3.4.1 cv::resize(INTER_LINEAR) for CV_8UC1 (SURVEY App.B H5) ---------------------- */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride)
{
    const int COEF_BITS = 11, COEF_SCALE = 1 << COEF_BITS;
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        int a0 = orc_cvround((double)(c0 * COEF_SCALE)), a1 = orc_cvround((double)(c1 * COEF_SCALE));
        ialpha[dx * 2] = (short)(a0 > 32767 ? 32767 : a0 < -32768 ? -32768 : a0);
        ialpha[dx * 2 + 1] = (short)(a1 > 32767 ? 32767 : a1 < -32768 ? -32768 : a1);
    }
    int* row0 = (int*)malloc(sizeof(int) * dw);
    int* row1 = (int*)malloc(sizeof(int) * dw);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        float c0 = 1.f - fy, c1 = fy;
        int b0 = orc_cvround((double)(c0 * COEF_SCALE)), b1 = orc_cvround((double)(c1 * COEF_SCALE));
        b0 = (short)(b0 > 32767 ? 32767 : b0 < -32768 ? -32768 : b0);
        b1 = (short)(b1 > 32767 ? 32767 : b1 < -32768 ? -32768 : b1);
        int sy0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);           /* clip(sy, 0, sh) */
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {                            /* HResizeLinear */
            int sx = xofs[dx];
            if (dx < xmax) {
                row0[dx] = S0[sx] * ialpha[dx * 2] + S0[sx + 1] * ialpha[dx * 2 + 1];
                row1[dx] = S1[sx] * ialpha[dx * 2] + S1[sx + 1] * ialpha[dx * 2 + 1];
            } else {
                row0[dx] = S0[sx] * COEF_SCALE;
                row1[dx] = S1[sx] * COEF_SCALE;
            }
        }
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)                              /* VResizeLinear 32s -> 8u */
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(row0); free(row1);
}

/* ---- GaussianBlur 8u: separable, Q8 fixed-point coefficients (SURVEY App.B H6) ----------------- */
void orc_gauss_kernel_q8(int n, double sigma, int* out)
{
    /* getGaussianKernel(n, sigma, CV_32F) then convertTo(CV_32S, 256) */
    float cf[32];
    double scale2X = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) {
        cf[i] = (float)(cf[i] * sum);
        out[i] = orc_cvround((double)cf[i] * 256.0);
    }
}

void orc_gaussian_blur5_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    int k[5];
    orc_gauss_kernel_q8(5, 2.0, k);
    int* tmp = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int t = -2; t <= 2; t++) acc += k[t + 2] * S[reflect101(x + t, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int t = -2; t <= 2; t++) acc += k[t + 2] * tmp[(size_t)reflect101(y + t, h) * w + x];
            int v = (acc + (1 << 15)) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(tmp);
}

/* ---- cv::FAST TYPE_9_16 with non-max suppression (OpenCV 3.4 fast.cpp; SURVEY App.B H7) ------- */
static const int k_ring[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}
};

static int corner_score16(const uint8_t* ptr, const int* pixel, int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int t = 4; t <= 8; t++) a = a < d[k + t] ? a : d[k + t];
        int m = a < d[k] ? a : d[k];
        a0 = a0 > m ? a0 : m;
        m = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > m ? a0 : m;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        b = b > d[k + 3] ? b : d[k + 3];
        for (int t = 4; t <= 5; t++) b = b > d[k + t] ? b : d[k + t];
        if (b >= b0) continue;
        for (int t = 6; t <= 8; t++) b = b > d[k + t] ? b : d[k + t];
        int m = b > d[k] ? b : d[k];
        b0 = b0 < m ? b0 : m;
        m = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < m ? b0 : m;
    }
    return -b0 - 1;
}

int orc_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, int* xys, int cap)
{
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = k_ring[k][0] + k_ring[k][1] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    if (w < 7 || h < 7) return 0;
    uint8_t* sbuf = (uint8_t*)calloc((size_t)3 * w, 1);
    int* cpbuf = (int*)malloc(sizeof(int) * 3 * (size_t)(w + 1));
    int count = 0;
    for (int i = 3; i < h - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        uint8_t* curr = sbuf + (size_t)((i - 3) % 3) * w;
        int* cornerpos = cpbuf + (size_t)((i - 3) % 3) * (w + 1) + 1;
        memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; j++, ptr++) {
                int v = ptr[0];
                int is_corner = 0;
                {   /* darker arc */
                    int vt = v - threshold, cnt = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) { if (++cnt > K) { is_corner = 1; break; } }
                        else cnt = 0;
                    }
                }
                if (!is_corner) {   /* brighter arc */
                    int vt = v + threshold, cnt = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) { if (++cnt > K) { is_corner = 1; break; } }
                        else cnt = 0;
                    }
                }
                if (is_corner) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = sbuf + (size_t)((i - 4 + 3) % 3) * w;
        const uint8_t* pprev = sbuf + (size_t)((i - 5 + 3) % 3) * w;
        cornerpos = cpbuf + (size_t)((i - 4 + 3) % 3) * (w + 1) + 1;
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] &&
                score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
                if (count < cap) { xys[count * 3] = j; xys[count * 3 + 1] = i - 1; xys[count * 3 + 2] = score; }
                count++;
            }
        }
    }
    free(sbuf); free(cpbuf);
    return count;
}

/* ---- octree: src/ORBextractor.cc:500-782 ---------------------------------------------------- */
typedef struct {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int* keys; int nkeys;
    int nomore;
    int prev, next;       /* std::list links (indices into the pool), -1 = none */
    int seq;              /* creation order: tie-break for the (size,pointer) sort (App.B H8) */
} onode;

typedef struct { onode* pool; int npool, cappool; int head, tail, size; int seqctr; } olist;

static int ol_new(olist* L)
{
    if (L->npool == L->cappool) {
        L->cappool = L->cappool ? L->cappool * 2 : 256;
        L->pool = (onode*)realloc(L->pool, sizeof(onode) * L->cappool);
    }
    onode* n = &L->pool[L->npool];
    memset(n, 0, sizeof(*n));
    n->prev = n->next = -1;
    n->seq = L->seqctr++;
    return L->npool++;
}
static void ol_push_front(olist* L, int id)
{
    L->pool[id].prev = -1; L->pool[id].next = L->head;
    if (L->head >= 0) L->pool[L->head].prev = id; else L->tail = id;
    L->head = id; L->size++;
}
static void ol_push_back(olist* L, int id)
{
    L->pool[id].next = -1; L->pool[id].prev = L->tail;
    if (L->tail >= 0) L->pool[L->tail].next = id; else L->head = id;
    L->tail = id; L->size++;
}
static int ol_erase(olist* L, int id)      /* returns next */
{
    int p = L->pool[id].prev, n = L->pool[id].next;
    if (p >= 0) L->pool[p].next = n; else L->head = n;
    if (n >= 0) L->pool[n].prev = p; else L->tail = p;
    L->size--;
    return n;
}

/* ExtractorNode::DivideNode :500-556; children ids returned in c[4] (not yet linked) */
static void divide_node(olist* L, int id, const orc_keypoint* kp, int c[4])
{
    for (int k = 0; k < 4; k++) c[k] = ol_new(L);           /* may realloc: take pointers after */
    onode* p = &L->pool[id];
    onode *n1 = &L->pool[c[0]], *n2 = &L->pool[c[1]], *n3 = &L->pool[c[2]], *n4 = &L->pool[c[3]];
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
    const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    n1->ULx = p->ULx; n1->ULy = p->ULy;
    n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
    n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;
    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = p->URx; n2->URy = p->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = p->URx; n2->BRy = p->ULy + halfY;
    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = p->BLx; n3->BLy = p->BLy;
    n3->BRx = n1->BRx; n3->BRy = p->BLy;
    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = p->BRx; n4->BRy = p->BRy;
    for (int k = 0; k < 4; k++) L->pool[c[k]].keys = (int*)malloc(sizeof(int) * (p->nkeys ? p->nkeys : 1));
    for (int i = 0; i < p->nkeys; i++) {
        const orc_keypoint* q = &kp[p->keys[i]];
        onode* t;
        if (q->x < (float)n1->URx) t = (q->y < (float)n1->BRy) ? n1 : n3;
        else t = (q->y < (float)n1->BRy) ? n2 : n4;
        t->keys[t->nkeys++] = p->keys[i];
    }
    for (int k = 0; k < 4; k++) if (L->pool[c[k]].nkeys == 1) L->pool[c[k]].nomore = 1;
}

typedef struct { int size; int seq; int id; } sizeptr;
static int cmp_sizeptr(const void* a, const void* b)
{
    const sizeptr* x = (const sizeptr*)a; const sizeptr* y = (const sizeptr*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

int orc_distribute_octree(const orc_keypoint* in, int n, int minX, int maxX, int minY, int maxY,
                          int N, orc_keypoint* out, int cap)
{
    olist L; memset(&L, 0, sizeof(L)); L.head = L.tail = -1;
    const int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni < 1) return -2;                                   /* reference divides by zero */
    const float hX = (float)(maxX - minX) / (float)nIni;
    int* ini = (int*)malloc(sizeof(int) * nIni);
    for (int i = 0; i < nIni; i++) {
        int id = ol_new(&L);
        onode* ni = &L.pool[id];
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        ni->keys = (int*)malloc(sizeof(int) * (n ? n : 1));
        ol_push_back(&L, id);
        ini[i] = id;
    }
    for (int i = 0; i < n; i++) {
        int b = (int)(in[i].x / hX);
        if (b < 0) b = 0; if (b >= nIni) b = nIni - 1;         /* reference would index out of range */
        onode* t = &L.pool[ini[b]];
        t->keys[t->nkeys++] = i;
    }
    free(ini);
    for (int lit = L.head; lit >= 0;) {
        onode* t = &L.pool[lit];
        if (t->nkeys == 1) { t->nomore = 1; lit = t->next; }
        else if (t->nkeys == 0) lit = ol_erase(&L, lit);
        else lit = t->next;
    }
    int finish = 0;
    sizeptr* vsp = NULL; int nvsp = 0, capvsp = 0;
#define VSP_PUSH(sz, idv) do { if (nvsp == capvsp) { capvsp = capvsp ? capvsp * 2 : 256; \
        vsp = (sizeptr*)realloc(vsp, sizeof(sizeptr) * capvsp); } \
        vsp[nvsp].size = (sz); vsp[nvsp].seq = L.pool[idv].seq; vsp[nvsp].id = (idv); nvsp++; } while (0)
    while (!finish) {
        int prevSize = L.size;
        int nToExpand = 0;
        nvsp = 0;
        for (int lit = L.head; lit >= 0;) {
            if (L.pool[lit].nomore) { lit = L.pool[lit].next; continue; }
            int c[4];
            divide_node(&L, lit, in, c);
            for (int k = 0; k < 4; k++) {
                if (L.pool[c[k]].nkeys > 0) {
                    ol_push_front(&L, c[k]);
                    if (L.pool[c[k]].nkeys > 1) { nToExpand++; VSP_PUSH(L.pool[c[k]].nkeys, c[k]); }
                }
            }
            lit = ol_erase(&L, lit);
        }
        if (L.size >= N || L.size == prevSize) {
            finish = 1;
        } else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                int nprev = nvsp;
                sizeptr* prev = (sizeptr*)malloc(sizeof(sizeptr) * (nprev ? nprev : 1));
                memcpy(prev, vsp, sizeof(sizeptr) * nprev);
                nvsp = 0;
                qsort(prev, nprev, sizeof(sizeptr), cmp_sizeptr);
                for (int j = nprev - 1; j >= 0; j--) {
                    int c[4];
                    divide_node(&L, prev[j].id, in, c);
                    for (int k = 0; k < 4; k++) {
                        if (L.pool[c[k]].nkeys > 0) {
                            ol_push_front(&L, c[k]);
                            if (L.pool[c[k]].nkeys > 1) VSP_PUSH(L.pool[c[k]].nkeys, c[k]);
                        }
                    }
                    ol_erase(&L, prev[j].id);
                    if (L.size >= N) break;
                }
                free(prev);
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
#undef VSP_PUSH
    int nout = 0;
    for (int lit = L.head; lit >= 0; lit = L.pool[lit].next) {
        onode* t = &L.pool[lit];
        int best = t->keys[0];
        float maxResponse = in[best].response;
        for (int k = 1; k < t->nkeys; k++)
            if (in[t->keys[k]].response > maxResponse) { best = t->keys[k]; maxResponse = in[best].response; }
        if (nout < cap) out[nout] = in[best];
        nout++;
    }
    for (int i = 0; i < L.npool; i++) free(L.pool[i].keys);
    free(L.pool); free(vsp);
    return nout;
}

/* ---- IC_Angle :77-104 ------------------------------------------------------------------------- */
float orc_ic_angle(const uint8_t* center, int step, const int* umax)
{
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ---- computeOrbDescriptor :108-157.  img: w x h continuous-with-step buffer WITHOUT border.
 * The reference indexes center[dy*step + dx] with no bounds check (SURVEY App.B H4): taps whose
 * linear index stays inside the buffer read the wrapped pixel exactly like the reference; taps
 * outside the allocation (reference: undefined behaviour) read 0 and the function returns 1. */
int orc_orb_descriptor(const uint8_t* img, int w, int h, int step, float kx, float ky, float angle_deg,
                       uint8_t* desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    float a = orc_cosf(angle), b = orc_sinf(angle);
    const int cx = orc_cvround((double)kx), cy = orc_cvround((double)ky);
    const long base = (long)cy * step + cx;
    const long total = (long)(h - 1) * step + w;     /* valid linear indices: [0, total) */
    int oob = 0;
    const signed char* pat = orc_bit_pattern_31;
    for (int i = 0; i < 32; i++, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            int t[2];
            for (int s = 0; s < 2; s++) {
                float px = (float)pat[(2 * k + s) * 2], py = (float)pat[(2 * k + s) * 2 + 1];
                float r0 = px * b, r1 = py * a;
                float c0 = px * a, c1 = py * b;
                int dy = orc_cvround((double)(r0 + r1));
                int dx = orc_cvround((double)(c0 - c1));
                long idx = base + (long)dy * step + dx;
                if (idx < 0 || idx >= total) { oob = 1; t[s] = 0; }
                else t[s] = img[idx];
            }
            val |= (t[0] < t[1]) << k;
        }
        desc[i] = (uint8_t)val;
    }
    return oob;
}

/* ---- ComputePyramid :1240-1265 ------------------------------------------------------------------ */
static void make_border(uint8_t* buf, int bw, int bh, int e, int w, int h)
{
    /* copyMakeBorder(REFLECT_101 [+ISOLATED]) of the centre ROI into the whole buffer */
    for (int y = 0; y < bh; y++) {
        int sy = reflect101(y - e, h);
        const uint8_t* S = buf + (size_t)(sy + e) * bw + e;
        uint8_t* D = buf + (size_t)y * bw;
        if (y - e >= 0 && y - e < h) {
            for (int x = 0; x < e; x++) D[x] = S[reflect101(x - e, w)];
            for (int x = e + w; x < bw; x++) D[x] = S[reflect101(x - e, w)];
        } else {
            for (int x = 0; x < bw; x++) D[x] = S[reflect101(x - e, w)];
        }
    }
}

static int compute_pyramid(orc_orb* e, const uint8_t* img, int W, int H, int stride)
{
    const int E = e->edge;
    for (int l = 0; l < e->p.nlevels; l++) {
        float scale = e->inv_sf[l];
        int w = orc_cvround((double)((float)W * scale)), h = orc_cvround((double)((float)H * scale));
        if (w < 1 || h < 1) return -2;
        e->lw[l] = w; e->lh[l] = h;
        int bw = w + 2 * E, bh = h + 2 * E;
        e->buf[l] = (uint8_t*)calloc((size_t)bw * bh, 1);
        uint8_t* roi = e->buf[l] + (size_t)E * bw + E;
        if (l != 0) {
            int pw = e->lw[l - 1], ph = e->lh[l - 1], pbw = pw + 2 * E;
            const uint8_t* proi = e->buf[l - 1] + (size_t)E * pbw + E;
            orc_resize_linear_u8(proi, pw, ph, pbw, roi, w, h, bw);
        } else {
            for (int y = 0; y < h; y++) memcpy(roi + (size_t)y * bw, img + (size_t)y * stride, w);
        }
        make_border(e->buf[l], bw, bh, E, w, h);
    }
    return 0;
}

/* ---- ComputeKeyPointsOctTree :784-902 ----------------------------------------------------------- */
static int compute_keypoints(orc_orb* e)
{
    const int E = e->edge;
    const int littleEdge = 3, littleEdgeX2 = 6;
    for (int level = 0; level < e->p.nlevels; ++level) {
        const int cols = e->lw[level], rows = e->lh[level], bw = cols + 2 * E;
        const uint8_t* roi = e->buf[level] + (size_t)E * bw + E;
        const int minBorderX = E - littleEdge, minBorderY = minBorderX;
        const int maxBorderX = cols - E + littleEdge, maxBorderY = rows - E + littleEdge;
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W_DENOM), nRows = (int)(height / W_DENOM);
        if (nCols < 1 || nRows < 1) return -2;                     /* SURVEY App.B H14 */
        const int wCell = (int)ceilf(width / (float)nCols), hCell = (int)ceilf(height / (float)nRows);
        int capc = ((cols + 1) / 2) * ((rows + 1) / 2) + 16;
        orc_keypoint* cand = (orc_keypoint*)malloc(sizeof(orc_keypoint) * capc);
        int* xys = (int*)malloc(sizeof(int) * 3 * capc);
        int nc = 0;
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + (float)hCell + (float)littleEdgeX2;
            if (iniY >= (float)(maxBorderY - littleEdge)) continue;
            if (maxY > (float)maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + (float)wCell + (float)littleEdgeX2;
                if (iniX >= (float)(maxBorderX - littleEdge)) continue;
                if (maxX > (float)maxBorderX) maxX = (float)maxBorderX;
                const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
                const uint8_t* sub = roi + (ptrdiff_t)y0 * bw + x0;
                int m = orc_fast9_16(sub, x1 - x0, y1 - y0, bw, e->p.iniThFAST, xys, capc);
                if (m == 0) m = orc_fast9_16(sub, x1 - x0, y1 - y0, bw, e->p.minThFAST, xys, capc);
                for (int k = 0; k < m; k++) {
                    orc_keypoint kp;
                    kp.x = (float)xys[k * 3] + (float)(j * wCell);
                    kp.y = (float)xys[k * 3 + 1] + (float)(i * hCell);
                    kp.size = 7.f; kp.angle = -1.f; kp.response = (float)xys[k * 3 + 2];
                    kp.octave = 0; kp.class_id = -1;
                    cand[nc++] = kp;
                }
            }
        }
        free(xys);
        e->cand[level] = cand; e->ncand[level] = nc;
        int capk = e->nfeat_level[level] + 64;                     /* result never exceeds N+2 (N>=4) */
        orc_keypoint* kps = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (capk > 0 ? capk : 1));
        int nk = 0;
        if (nc > 0) {
            nk = orc_distribute_octree(cand, nc, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                       e->nfeat_level[level], kps, capk);
            if (nk < 0) { free(kps); return -2; }
            if (nk > capk) { free(kps); return -3; }
        }
        const int scaledPatchSize = (int)((float)PATCH_SIZE * e->sf[level]);
        for (int i = 0; i < nk; i++) {
            kps[i].x += (float)minBorderX;
            kps[i].y += (float)minBorderY;
            kps[i].octave = level;
            kps[i].size = (float)scaledPatchSize;
        }
        e->kps[level] = kps; e->nkps[level] = nk;
    }
    for (int level = 0; level < e->p.nlevels; ++level) {           /* computeOrientation :491-498 */
        const int bw = e->lw[level] + 2 * E;
        const uint8_t* roi = e->buf[level] + (size_t)E * bw + E;
        for (int i = 0; i < e->nkps[level]; i++) {
            orc_keypoint* kp = &e->kps[level][i];
            const uint8_t* center = roi + (size_t)orc_cvround((double)kp->y) * bw + orc_cvround((double)kp->x);
            kp->angle = orc_ic_angle(center, bw, e->umax);
        }
    }
    return 0;
}

/* ---- operator() :1092-1238 ------------------------------------------------------------------------ */
int orc_orb_extract(orc_orb* e, const uint8_t* img, int W, int H, int stride, int lap0, int lap1,
                    int want_desc, orc_keypoint* out, uint8_t* desc, uint8_t* oob, int cap, int* n_out)
{
    if (n_out) *n_out = 0;
    if (!img || W <= 0 || H <= 0) return -1;                     /* _image.empty() :1096 */
    free_state(e);
    int rc = compute_pyramid(e, img, W, H, stride);
    if (rc) return rc;
    rc = compute_keypoints(e);
    if (rc) return rc;
    int nkeypoints = 0;
    for (int l = 0; l < e->p.nlevels; l++) nkeypoints += e->nkps[l];
    if (nkeypoints > cap) return -3;
    if (n_out) *n_out = nkeypoints;
    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    uint8_t* dtmp = (uint8_t*)malloc(32 * (size_t)(nkeypoints ? nkeypoints : 1));
    const int E = e->edge;
    for (int level = 0; level < e->p.nlevels; ++level) {
        int nl = e->nkps[level];
        if (nl == 0) continue;
        const int w = e->lw[level], h = e->lh[level], bw = w + 2 * E;
        uint8_t* ooblev = (uint8_t*)calloc(nl, 1);
        if (want_desc) {
            /* workingMat = mvImagePyramid[level].clone(); GaussianBlur(5x5, 2, 2, REFLECT_101) */
            const uint8_t* roi = e->buf[level] + (size_t)E * bw + E;
            e->blur[level] = (uint8_t*)malloc((size_t)w * h);
            orc_gaussian_blur5_u8(roi, w, h, bw, e->blur[level], w);
            for (int i = 0; i < nl; i++)
                ooblev[i] = (uint8_t)orc_orb_descriptor(e->blur[level], w, h, w, e->kps[level][i].x,
                                                        e->kps[level][i].y, e->kps[level][i].angle,
                                                        dtmp + 32 * (size_t)i);
        }
        float scale = e->sf[level];
        for (int i = 0; i < nl; i++) {
            orc_keypoint kp = e->kps[level][i];
            if (level != 0) { kp.x *= scale; kp.y *= scale; }
            int dst;
            if (kp.x >= (float)lap0 && kp.x <= (float)lap1) dst = stereoIndex--;
            else dst = monoIndex++;
            out[dst] = kp;
            if (want_desc && desc) memcpy(desc + 32 * (size_t)dst, dtmp + 32 * (size_t)i, 32);
            if (oob) oob[dst] = ooblev[i];
        }
        free(ooblev);
    }
    free(dtmp);
    return monoIndex;
}

int orc_orb_level_size(const orc_orb* e, int level, int* w, int* h)
{ if (level < 0 || level >= e->p.nlevels) return -1; *w = e->lw[level]; *h = e->lh[level]; return 0; }
const uint8_t* orc_orb_level_buffer(const orc_orb* e, int level, int* bw, int* bh)
{ *bw = e->lw[level] + 2 * e->edge; *bh = e->lh[level] + 2 * e->edge; return e->buf[level]; }
const uint8_t* orc_orb_level_blur(const orc_orb* e, int level) { return e->blur[level]; }
int orc_orb_level_candidates(const orc_orb* e, int level, const orc_keypoint** out)
{ *out = e->cand[level]; return e->ncand[level]; }
int orc_orb_level_keypoints(const orc_orb* e, int level, const orc_keypoint** out)
{ *out = e->kps[level]; return e->nkps[level]; }

/* ---- tracked keypoints: ComputeTrackedKPtsDesc :1316-1363, AssignKPtLevelByBestDesc :1267-1314 ------------------- */
static int pyramid_and_blur(orc_orb* e, const uint8_t* img, int W, int H, int stride)
{
    free_state(e);
    int rc = compute_pyramid(e, img, W, H, stride);
    if (rc) return rc;
    const int E = e->edge;
    for (int l = 0; l < e->p.nlevels; l++) {
        const int w = e->lw[l], h = e->lh[l], bw = w + 2 * E;
        const uint8_t* roi = e->buf[l] + (size_t)E * bw + E;
        e->blur[l] = (uint8_t*)malloc((size_t)w * h);
        orc_gaussian_blur5_u8(roi, w, h, bw, e->blur[l], w);
    }
    return 0;
}

int orc_orb_tracked_descriptors(orc_orb* e, const uint8_t* img, int W, int H, int stride, const orc_keypoint* kps, int n,
                                uint8_t* desc, uint8_t* oob)
{
    if (!img || W <= 0 || H <= 0) return -1;                     /* trackedImage.empty() -> return */
    int rc = pyramid_and_blur(e, img, W, H, stride);
    if (rc) return rc;
    memset(desc, 0, 32 * (size_t)n);
    if (oob) memset(oob, 0, n);
    for (int level = 0; level < e->p.nlevels; level++) {
        const float scale = e->inv_sf[level];
        for (int i = 0; i < n; i++) {
            if (kps[i].octave != level) continue;
            const float x = kps[i].x * scale, y = kps[i].y * scale;      /* currKPt.pt = currKPt.pt * scale */
            int o = orc_orb_descriptor(e->blur[level], e->lw[level], e->lh[level], e->lw[level], x, y, kps[i].angle,
                                       desc + 32 * (size_t)i);
            if (oob) oob[i] = (uint8_t)o;
        }
    }
    return 0;
}

static int hamming32(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int i = 0; i < 32; i++) { unsigned v = a[i] ^ b[i]; while (v) { d += v & 1; v >>= 1; } }
    return d;
}

int orc_orb_assign_level_by_best_desc(orc_orb* e, const uint8_t* img, int W, int H, int stride, const uint8_t* ref_desc,
                                      orc_keypoint* kps, int n)
{
    if (!img || W <= 0 || H <= 0) return -1;
    int rc = pyramid_and_blur(e, img, W, H, stride);
    if (rc) return rc;
    int* minDist = (int*)malloc(sizeof(int) * (n ? n : 1));
    for (int i = 0; i < n; i++) minDist[i] = 0x7fffffff;
    for (int level = 0; level < e->p.nlevels; level++) {
        const float scale = e->inv_sf[level];
        for (int i = 0; i < n; i++) {
            uint8_t d[32];
            const float x = kps[i].x * scale, y = kps[i].y * scale;
            orc_orb_descriptor(e->blur[level], e->lw[level], e->lh[level], e->lw[level], x, y, kps[i].angle, d);
            const int dist = hamming32(ref_desc + 32 * (size_t)i, d);    /* ORBmatcher::DescriptorDistance */
            if (dist < minDist[i]) { minDist[i] = dist; kps[i].octave = level; }
        }
    }
    free(minDist);
    return 0;
}
