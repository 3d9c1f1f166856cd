// gbp_fused.hpp -- fused single-kernel sweep (placeholder: plan never enables; general path runs)
#pragma once
#include "gbp_kernels.hpp"
#include <vector>
#include <cstdint>

namespace gbp {

struct FusedPlan {
    bool enabled = false;
    int n_tiles = 0, n_blocks = 0;
};

inline int fused_plan(FusedPlan &, const Params &, const std::vector<int32_t> &, const std::vector<int32_t> &,
                      hipStream_t, int) { return 0; }
inline int fused_launch(FusedPlan &, const Params &, int, int, double *, hipStream_t) { return 0; }
inline void fused_destroy(FusedPlan &) {}

}  // namespace gbp
