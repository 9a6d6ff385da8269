// tsp_pipeline.hip -- the three-class splat pipeline (default render path).
#include <vector>

#include "tsp_internal.h"

namespace tsp {

int launch_pipeline(tsp_context *ctx, const Camera &cam, const int64_t *h_starts, const int64_t *h_lens, int n_ranges,
                    int64_t total, int mode) {
    // interim: route through the generic kernel until the class kernels land
    std::vector<int64_t> pack(3 * (size_t)n_ranges + 1);
    int64_t acc = 0;
    for (int i = 0; i < n_ranges; ++i) {
        pack[i] = h_starts[i];
        pack[n_ranges + i] = h_lens[i];
        pack[2 * n_ranges + i] = acc;
        acc += h_lens[i];
    }
    pack[3 * n_ranges] = acc;
    if (ctx->ws.range_capacity < (int64_t)pack.size()) {
        if (ctx->ws.range_prefix) TSP_HIP(hipFree(ctx->ws.range_prefix));
        ctx->ws.range_capacity = (int64_t)pack.size() * 2 + 64;
        TSP_HIP(hipMalloc((void **)&ctx->ws.range_prefix, ctx->ws.range_capacity * sizeof(int64_t)));
    }
    TSP_HIP(hipMemcpyAsync(ctx->ws.range_prefix, pack.data(), pack.size() * sizeof(int64_t), hipMemcpyHostToDevice,
                           ctx->stream));
    int rc = launch_generic(ctx, cam, ctx->ws.range_prefix, n_ranges, total, mode);
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    return rc;
}

}  // namespace tsp
