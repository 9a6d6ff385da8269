// tsp_group.hip -- several GPUs of one node behind ONE handle, for C clients (SURVEY.md section 8b sketched
// `tsp_create(n_devices, device_ids, ...)`; section 8e).
//
// A group is G ordinary contexts (one per device) plus the host-thread choreography a caller would otherwise write
// himself: particles are cut into the contiguous index ranges [g N / G, (g + 1) N / G) (the arithmetic of the reference's
// SplitBuffers._calculate_splits, src/topsy/split_buffers.py:26-38), a render block's (start, len) ranges are intersected
// with every shard (global_to_split_monotonic, :78-116) and the G tsp_render calls run concurrently, one host thread per
// context; a frame ends with tsp_group_end_frame = the ONE sum-reduce of the float32 image onto context 0 (RCCL over xGMI,
// or -- when two contexts share a device, which RCCL refuses: single-GPU test boxes, or when librccl cannot be loaded -- a
// read-back / add / write-back of the float32 presentation images through the host).  Everything that looks at the finished frame (tsp_read_image, tsp_colormap_*, tsp_content_*,
// tsp_tile_periodic) is called on tsp_group_context(group, 0).  The Python layer's MultiGpuContext (topsy_amd/multigpu.py)
// is the same choreography in Python (it also offers the block-cyclic assignment for cell-sorted loaders).
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "tsp_internal.h"

struct tsp_group {
    std::vector<tsp_context *> ctx;
    std::vector<int64_t> bounds;          // shard g owns global indices [bounds[g], bounds[g + 1])
    bool rccl = false;                    // distinct devices: RCCL; else the host collective
    bool needs_reduce = false;            // partial images were rendered since the last end_frame
    int R = 0, Ccap = 0;
};

namespace {

using tsp::set_error;

// run fn(g) on one host thread per context; first failure wins (its code, its thread's error text)
template <typename F> int for_each_context(tsp_group *grp, F fn) {
    const int G = (int)grp->ctx.size();
    std::vector<int> rc(G, TSP_OK);
    std::vector<std::string> msg(G);
    std::vector<std::thread> th;
    th.reserve(G);
    for (int g = 0; g < G; ++g)
        th.emplace_back([&, g]() {
            rc[g] = fn(g);
            if (rc[g] != TSP_OK) msg[g] = tsp_last_error();      // tsp_last_error is thread-local: copy it out
        });
    for (auto &t : th) t.join();
    for (int g = 0; g < G; ++g)
        if (rc[g] != TSP_OK) {
            set_error("context %d (device %d): %s", g, grp->ctx[g]->device, msg[g].c_str());
            return rc[g];
        }
    return TSP_OK;
}

void set_bounds(tsp_group *grp, int64_t n) {
    const int G = (int)grp->ctx.size();
    grp->bounds.assign(G + 1, 0);
    for (int g = 0; g <= G; ++g) grp->bounds[g] = (n * g) / G;
}

}  // namespace

extern "C" {

int tsp_group_create(int n_devices, const int *device_ids, int resolution, int n_channels, tsp_group **out) {
    TSP_REQUIRE(out && device_ids, TSP_EINVAL, "NULL argument");
    *out = nullptr;
    TSP_REQUIRE(n_devices >= 1 && n_devices <= 64, TSP_EINVAL, "n_devices %d out of range", n_devices);
    tsp_group *grp = new tsp_group();
    grp->R = resolution;
    grp->Ccap = n_channels;
    for (int g = 0; g < n_devices; ++g) {
        tsp_context *c = nullptr;
        const int rc = tsp_create(device_ids[g], resolution, n_channels, &c);
        if (rc) {
            for (tsp_context *p : grp->ctx) tsp_destroy(p);
            delete grp;
            return rc;
        }
        grp->ctx.push_back(c);
    }
    std::vector<int> ids(device_ids, device_ids + n_devices);
    std::sort(ids.begin(), ids.end());
    grp->rccl = n_devices > 1 && std::adjacent_find(ids.begin(), ids.end()) == ids.end();
    set_bounds(grp, 0);
    if (grp->rccl) {
        char uid[TSP_UNIQUE_ID_BYTES];
        if (tsp_comm_unique_id(uid) != TSP_OK) {
            // RCCL cannot be loaded: the sum still exists -- through the host (what contexts that share a device always use)
            grp->rccl = false;
        } else {
            // ncclCommInitRank blocks until every rank has joined: all G calls at once.  Whatever can fail BEFORE the collective
            // (selecting the device: the tsp_create calls above already ran on each of them) has been checked on this thread, so
            // no rank can drop out while the others wait inside the collective.
            int rc = TSP_OK;
            for (int g = 0; g < n_devices && !rc; ++g)
                if (hipSetDevice(grp->ctx[g]->device) != hipSuccess) { set_error("cannot select device %d", grp->ctx[g]->device); rc = TSP_EHIP; }
            if (!rc) rc = for_each_context(grp, [&](int g) { return tsp_comm_init(grp->ctx[g], n_devices, g, uid); });
            if (rc) {
                const std::string keep = tsp_last_error();
                for (tsp_context *p : grp->ctx) tsp_destroy(p);       // (tsp_destroy tears down whatever communicator exists)
                delete grp;
                set_error("%s", keep.c_str());
                return rc;
            }
        }
    }
    *out = grp;
    return TSP_OK;
}

void tsp_group_destroy(tsp_group *grp) {
    if (!grp) return;
    for (tsp_context *c : grp->ctx) tsp_comm_destroy(c);      // the communicators first, rank by rank
    for (tsp_context *c : grp->ctx) tsp_destroy(c);
    delete grp;
}

int tsp_group_size(tsp_group *grp) { return grp ? (int)grp->ctx.size() : 0; }

tsp_context *tsp_group_context(tsp_group *grp, int index) {
    return (grp && index >= 0 && index < (int)grp->ctx.size()) ? grp->ctx[index] : nullptr;
}

int tsp_group_uses_rccl(tsp_group *grp) { return grp && grp->rccl ? 1 : 0; }

int tsp_group_set_kernel_mips(tsp_group *grp, const float *lut, int n0, int n_levels) {
    TSP_REQUIRE(grp, TSP_EINVAL, "NULL group");
    return for_each_context(grp, [&](int g) { return tsp_set_kernel_mips(grp->ctx[g], lut, n0, n_levels); });
}

int tsp_group_upload_particles(tsp_group *grp, int64_t n, const float *x, const float *y, const float *z, const float *h,
                               const float *mass) {
    TSP_REQUIRE(grp && x && y && z && h, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(n >= 0, TSP_EINVAL, "negative particle count");
    set_bounds(grp, n);
    return for_each_context(grp, [&](int g) {
        const int64_t a = grp->bounds[g], len = grp->bounds[g + 1] - a;
        return tsp_upload_particles(grp->ctx[g], len, x + a, y + a, z + a, h + a, mass ? mass + a : nullptr);
    });
}

int tsp_group_upload_quantity(tsp_group *grp, const float *q) {
    TSP_REQUIRE(grp, TSP_EINVAL, "NULL group");
    return for_each_context(grp, [&](int g) { return tsp_upload_quantity(grp->ctx[g], q ? q + grp->bounds[g] : nullptr); });
}

int tsp_group_upload_rgb(tsp_group *grp, const float *r, const float *g_, const float *b) {
    TSP_REQUIRE(grp && r && g_ && b, TSP_EINVAL, "NULL argument");
    return for_each_context(grp, [&](int g) {
        const int64_t a = grp->bounds[g];
        return tsp_upload_rgb(grp->ctx[g], r + a, g_ + a, b + a);
    });
}

int tsp_group_generate_synthetic(tsp_group *grp, int64_t n_total, int64_t first, int64_t count, uint64_t seed, float h_cap,
                                 int with_quantity, int with_rgb) {
    TSP_REQUIRE(grp, TSP_EINVAL, "NULL group");
    TSP_REQUIRE(count >= 0, TSP_EINVAL, "negative particle count");
    set_bounds(grp, count);
    return for_each_context(grp, [&](int g) {
        const int64_t a = grp->bounds[g], len = grp->bounds[g + 1] - a;
        return tsp_generate_synthetic(grp->ctx[g], n_total, first + a, len, seed, h_cap, with_quantity, with_rgb);
    });
}

int tsp_group_reorder_spatial(tsp_group *grp, int n_strata, uint64_t seed) {
    TSP_REQUIRE(grp, TSP_EINVAL, "NULL group");
    const int G = (int)grp->ctx.size();
    const int per = std::max(1, (n_strata + G - 1) / G);      // the snapshot as a whole keeps about n_strata block boundaries
    return for_each_context(grp, [&](int g) {
        if (grp->bounds[g + 1] == grp->bounds[g]) return (int)TSP_OK;
        return tsp_reorder_spatial(grp->ctx[g], per, seed, nullptr);
    });
}

int64_t tsp_group_num_particles(tsp_group *grp) { return grp ? grp->bounds.back() : 0; }

int tsp_group_set_option(tsp_group *grp, const char *name, int64_t value) {
    TSP_REQUIRE(grp && name, TSP_EINVAL, "NULL argument");
    return for_each_context(grp, [&](int g) { return tsp_set_option(grp->ctx[g], name, value); });
}

int tsp_group_render(tsp_group *grp, const float *M, float scale_factor, const int64_t *starts, const int64_t *lens, int n_ranges,
                     int clear, int mode, int flags, double *gpu_ms_out) {
    TSP_REQUIRE(grp && M, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(n_ranges >= 0 && (n_ranges == 0 || (starts && lens) || (!starts && !lens)), TSP_EINVAL, "bad ranges");
    const int G = (int)grp->ctx.size();
    const int64_t all_start = 0, all_len = grp->bounds.back();
    if (!starts) { starts = &all_start; lens = &all_len; n_ranges = 1; }
    for (int i = 0; i < n_ranges; ++i) TSP_REQUIRE(lens[i] >= 0, TSP_EINVAL, "range %d has negative length %lld", i, (long long)lens[i]);
    std::vector<double> ms(G, 0.0);
    const int rc = for_each_context(grp, [&](int g) {
        // clip the global ranges to shard g and re-base them; a shard the block does not touch still takes part with an
        // explicit empty range (clear must reach every partial image)
        const int64_t a = grp->bounds[g], b = grp->bounds[g + 1];
        std::vector<int64_t> s, l;
        const int64_t n = grp->bounds.back();
        for (int i = 0; i < n_ranges; ++i) {
            // first to [0, n) as tsp_render does (a caller may pass INT64_MAX for "to the end"), then to the shard
            int64_t r0 = starts[i], len = lens[i];
            if (r0 < 0) { len = (len > -r0) ? len + r0 : 0; r0 = 0; }
            if (r0 >= n || len == 0) continue;
            if (len > n - r0) len = n - r0;
            const int64_t lo = std::max(r0, a), hi = std::min(r0 + len, b);
            if (hi > lo) { s.push_back(lo - a); l.push_back(hi - lo); }
        }
        if (s.empty()) { s.push_back(0); l.push_back(0); }
        return tsp_render(grp->ctx[g], M, scale_factor, s.data(), l.data(), (int)s.size(), clear, mode, flags, &ms[g]);
    });
    if (rc) return rc;
    grp->needs_reduce = true;
    if (gpu_ms_out) *gpu_ms_out = *std::max_element(ms.begin(), ms.end());
    return TSP_OK;
}

int tsp_group_end_frame(tsp_group *grp, double *ms_out) {
    TSP_REQUIRE(grp, TSP_EINVAL, "NULL group");
    if (ms_out) *ms_out = 0.0;
    if (!grp->needs_reduce || grp->ctx.size() == 1) { grp->needs_reduce = false; return TSP_OK; }
    const int G = (int)grp->ctx.size();
    // Every member's preconditions are checked HERE, on the calling thread: a rank that failed on its own thread after the
    // others had entered the collective would leave them waiting in it for ever (a caller may have touched one member through
    // tsp_group_context, e.g. rendered another mode on it or reduced it by hand)
    for (int g = 0; g < G; ++g) {
        const tsp_context *c = grp->ctx[g];
        TSP_REQUIRE(c->C == grp->ctx[0]->C, TSP_ESTATE, "context %d holds a %d-channel image, context 0 a %d-channel one", g, c->C, grp->ctx[0]->C);
        TSP_REQUIRE(!c->image_is_reduced, TSP_ESTATE, "context %d was already reduced for this frame", g);
        TSP_REQUIRE(!grp->rccl || c->comm, TSP_ESTATE, "context %d has lost its communicator", g);
    }
    grp->needs_reduce = false;
    if (grp->rccl) {
        std::vector<double> ms(G, 0.0);
        const int rc = for_each_context(grp, [&](int g) { return tsp_comm_reduce_image(grp->ctx[g], 0, &ms[g]); });
        if (rc) return rc;
        if (ms_out) *ms_out = *std::max_element(ms.begin(), ms.end());
        return TSP_OK;
    }
    // Host collective (contexts that share a device, which RCCL refuses -- single-GPU test boxes -- or no RCCL at all): the
    // float32 partial images are read back, added in rank order and the sum becomes context 0's PRESENTATION image only, exactly
    // what the RCCL reduce does in place: every float64 accumulator, the root's included, stays shard-local, so a later
    // tsp_group_render with clear = 0 continues from unrounded partial sums (the REFINE contract of include/topsy_splat.h)
    const size_t count = (size_t)grp->R * grp->R * grp->ctx[0]->C;
    std::vector<std::vector<float>> part(G, std::vector<float>(count));
    int rc = for_each_context(grp, [&](int g) { return tsp_read_image(grp->ctx[g], part[g].data()); });
    if (rc) return rc;
    std::vector<float> sum(part[0]);
    for (int g = 1; g < G; ++g)
        for (size_t i = 0; i < count; ++i) sum[i] += part[g][i];         // float32, rank order: as ncclReduce(sum, float32)
    return tsp_set_reduced_image(grp->ctx[0], sum.data());
}

int tsp_group_shard_range(tsp_group *grp, int index, int64_t *first_out, int64_t *count_out) {
    TSP_REQUIRE(grp && first_out && count_out, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(index >= 0 && index < (int)grp->ctx.size(), TSP_EINVAL, "shard %d of %d", index, (int)grp->ctx.size());
    *first_out = grp->bounds[index];
    *count_out = grp->bounds[index + 1] - grp->bounds[index];
    return TSP_OK;
}

int tsp_group_upload_band_magnitudes(tsp_group *grp, int n_bands, const double *mags, const double *weights) {
    TSP_REQUIRE(grp && mags && weights, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(n_bands >= 1 && n_bands <= 64, TSP_EINVAL, "n_bands %d out of range", n_bands);
    // mags is [n_bands][n] over the WHOLE snapshot: every shard takes its columns of every band
    const int64_t n = grp->bounds.back();
    return for_each_context(grp, [&](int g) {
        const int64_t a = grp->bounds[g], len = grp->bounds[g + 1] - a;
        if (len == 0) return (int)TSP_OK;
        std::vector<double> cut((size_t)n_bands * len);
        for (int k = 0; k < n_bands; ++k) std::copy(mags + (size_t)k * n + a, mags + (size_t)k * n + a + len, cut.begin() + (size_t)k * len);
        return tsp_upload_band_magnitudes(grp->ctx[g], n_bands, cut.data(), weights);
    });
}

int tsp_group_get_stats(tsp_group *grp, tsp_stats *out) {
    TSP_REQUIRE(grp && out, TSP_EINVAL, "NULL argument");
    tsp_stats t = {};
    for (tsp_context *c : grp->ctx) {
        tsp_stats s;
        const int rc = tsp_get_stats(c, &s);
        if (rc) return rc;
        // counters add up over the shards, times are the slowest shard's
        t.n_particles += s.n_particles; t.n_small += s.n_small; t.n_mid += s.n_mid; t.n_huge += s.n_huge; t.n_culled += s.n_culled;
        t.n_fragments += s.n_fragments; t.n_mega += s.n_mega;
        t.n_fragments_stream += s.n_fragments_stream; t.n_fragments_mid += s.n_fragments_mid;
        t.n_fragments_huge += s.n_fragments_huge; t.n_fragments_mega += s.n_fragments_mega; t.n_chunk_culled += s.n_chunk_culled;
        t.ms_stream = std::max(t.ms_stream, s.ms_stream); t.ms_mid = std::max(t.ms_mid, s.ms_mid);
        t.ms_huge = std::max(t.ms_huge, s.ms_huge); t.ms_mega = std::max(t.ms_mega, s.ms_mega);
        t.ms_total = std::max(t.ms_total, s.ms_total);
    }
    *out = t;
    return TSP_OK;
}

}  // extern "C"
