// Pieces of the ball query shared by the brute-force kernel (grouping.hip) and the spatially
// culled one (nn_sorted.hip).
#pragma once
#include "common.h"

namespace prcnn {

constexpr int BQ_MAX_SCALES = 4;
struct BallScales {            // up to BQ_MAX_SCALES radii evaluated in one pass (MSG grouping)
    float thr2[4];
    int nsample[4];
    int *idx[4];
    int *pts_cnt[4];
};

// smallest float t such that max(sqrtf(t), 1e-20f) >= radius (defined in grouping.hip)
float ball_threshold(float radius);

}  // namespace prcnn
