"""Constants of the render path.  Values are the reference's (src/topsy/config.py:1-44) so that
time-budgeted progressive rendering and the colormap LUT behave identically."""

DEFAULT_RESOLUTION = 1024            # config.py:1
DEFAULT_COLORMAP = "twilight_shifted"  # config.py:2
DEFAULT_SCALE = 200.0                # config.py:4  half-width of the view, kpc
TARGET_FPS = 30                      # config.py:6
INITIAL_PARTICLES_TO_RENDER = 1e5    # config.py:7
COLORMAP_NUM_SAMPLES = 1000          # config.py:14
TEST_DATA_NUM_PARTICLES_DEFAULT = int(1e6)   # config.py:16
# The reference cuts an EXPORT frame into submissions of 2^25 particles (config.py:22-25: "pipeline stalls" of a WebGPU render
# pass above that size).  A HIP launch has no such limit -- the 1e9-particle snapshot renders in ONE tsp_render call -- and every
# extra block costs the tile kernels' per-launch start-up again (30 blocks: ~3x the frame time at 1e9), so this backend's cap
# is "never" (tsp_render itself takes < 2^32 particles per context).  The reference's value stays selectable:
# tests replay the reference's golden block sequences with it (tests/test_host_logic.py).
REFERENCE_MAX_PARTICLES_PER_EXPORT_RENDERCALL = 2 ** 25   # config.py:22
MAX_PARTICLES_PER_EXPORT_RENDERCALL = 2 ** 40
DEFAULT_CELLS_NSIDE = 16             # config.py:27
CELL_LAYOUT_FRACTIONAL_PADDING = 1e-5  # config.py:33

# --- backend-specific knobs (no reference counterpart) -------------------------------------
# smallest number of uniform random strata used by the load-time spatial ordering (tsp_reorder_spatial):
# index prefixes stay unbiased samples at 1/STRATA granularity while runs stay screen-coherent.  Fewer strata = denser strata =
# more compact 512-particle chunks on screen (round 5, 32 -> 8: one shard of the 1e9 snapshot 10.15 -> 9.78 ms, exactly 1e8
# particles 18.10 -> 17.94, 1.25e8 20.3 -> 19.9; <= 1e7 particles unchanged); a stratum of 1/8 of a <= 2.5e8-particle snapshot
# still renders in 1-3 ms, well inside the 1/30 s budget of the progressive renderer, whose smallest block it is.
SPATIAL_ORDER_STRATA = 8
# large snapshots get more strata so that a stratum -- the smallest spatially unbiased block of the progressive
# renderer -- holds at most about this many particles: 3.2e7 particles render in ~6 ms on an MI355X, well inside the
# 1/30 s frame budget.  Fewer, larger strata keep 512-particle chunks more local on screen: the 1e9-particle snapshot in
# 32 strata renders in 65.8 ms, in 250 (the 4e6-particle strata of rounds 1-4) 67.4 ms (the mid-footprint kernel 16.2 / 17.6 ms); 16 cost
# kernel S 1.2 ms (more same-pixel collisions in its LDS window)
MAX_PARTICLES_PER_STRATUM = 32_000_000
SPATIAL_ORDER_MAX_STRATA = 400
# Several GPUs behind one Visualizer (multigpu.py): how the caller's particle order is cut into shards.  "contiguous" =
# index ranges [g N / G, (g + 1) N / G) (split_buffers.py:26-38); "interleaved" = blocks of MULTI_GPU_INTERLEAVE_BLOCK
# consecutive particles dealt to the shards in turn; "auto" = interleaved when the loader brings its own cell layout (its
# order is then spatially sorted and index ranges are spatial slabs of very different cost), contiguous otherwise.
MULTI_GPU_SHARD_ASSIGNMENT = "auto"
MULTI_GPU_INTERLEAVE_BLOCK = 4096
