// Argument blocks of the KeyFrame-side matchers, shared by match.hip (kernels) and eorb_fe.hip (C ABI).
#pragma once
#include <stdint.h>
#include "../../include/eorb_fe.h"

namespace eorb {

struct GridB { float minX, minY, invW, invH; };

struct TriArgs {
    const eorb_keypoint* kps1; int n1; const uint8_t* desc1; int stride1; const uint8_t* elig1;
    const uint32_t* nodes1; const int32_t* off1; const int32_t* idx1; int nn1;
    const eorb_keypoint* kps2; int n2; const uint8_t* desc2; int stride2; const uint8_t* elig2;
    const uint32_t* nodes2; const int32_t* off2; const int32_t* idx2; int nn2;
    float epx, epy; float F[9]; const float* scale2; const float* sigma2_2; int nlevels;
    int bCoarse, checkOri;
    int32_t* match12; int8_t* bin1; int32_t* histo; int32_t* nmatches;
};

struct RadArgs {
    const eorb_keypoint* kps; int n; const uint8_t* desc; int stride; GridB g;
    const uint16_t* cell;                           // n: ix*48+iy or 0xFFFF (Frame::PosInGrid), from kf_cells_kernel
    int M; const uint8_t* valid; const float* uv; const float* radius; const int32_t* level; const uint8_t* q_desc;
    const float* inv_sigma2; int nlevels;
    const float* uright; const float* q_ur;       // Fuse, rectified stereo: mvuRight per keypoint (>= 0: has one), the map points' predicted right coordinates
    uint8_t* taken; float accept_thr;
    int32_t* best_idx; int32_t* best_dist;
};

// DBoW2 vocabulary tree, device resident (TemplatedVocabulary::m_nodes flattened; node 0 = root)
struct BowVoc {
    int nnodes, L;
    const int32_t* child_off; const int32_t* child_ids; const uint8_t* node_desc; const int32_t* word_id; const double* weight;
};

}  // namespace eorb
