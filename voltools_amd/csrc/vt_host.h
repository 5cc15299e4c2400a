// vt_host.h -- host-side declarations shared by the C ABI (vt_api.hip) and the launch planner (vt_plan.hip).
#pragma once
#include "vt_internal.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

// Experiment overrides, read from the environment once per handle (vt_volume_create) -- never on the per-call path.
// None is needed in production; the A/B runs quoted in DESIGN.md (tools/march_ab.py --env) and a few parity tests use them to reach
// planner alternatives.  read() below is the list: the knobs outside its #ifdef blocks are read by the product build, the others by the
// test build (VT_LEGACY: round 1's kernels and the settled A/Bs of round 3) or the ablation build (VT_EXPERIMENTS).
struct Tuning {
    int tile = -1;                 // VT_TILE: force tile configuration (index into the kernel family's table)
    int la = 0;                    // VT_LA: planes / pairs staged ahead by the marching kernels
    int march_box = -1;            // VT_MARCH_BOX: 1 = bounding-box staging, 0 = packed row spans, -1 = planner's choice
    int lxpad = -1;                // VT_LXPAD: LDS row padding of the pair kernel's boxes
    int dch = 0;                   // VT_DCH: output planes per marching chunk
    int blk_h = -1, blk_w = -1;    // VT_BLK_H / VT_BLK_W: blocked tile order of the marching kernels (tiles per block; 0 = plain order)
    bool plain_tile_order = false; // VT_TILE_ORDER=0: packed kernel walks tiles in plain instead of blocked order
    int rswap_wfast = -1;          // VT_RSWAP_WFAST: 1 / 0 = w-fastest / h-fastest tile order on the in-plane transposed copy, -1 = by kernel
    bool exp_nostore = false;      // VT_EXP_NOSTORE / VT_EXP_NOLOAD / VT_EXP_NOLDS: ablation builds (-DVT_EXPERIMENTS) only
    bool exp_noload = false;
    bool exp_nolds = false;
    bool exp_noloop = false;
    bool exp_stamps = false;       // VT_EXP_STAMPS (ablation build): phase timers of the span kernel, written over the first floats of the output
    bool exp_notiles = false;      // VT_EXP_NOTILES (ablation build): the span kernel's workgroups do their set-up and leave
    bool exp_static = false;       // VT_EXP_STATIC (ablation build): persistent kernels stride through their tile ids instead of fetching them from the queue
    int quad_rows = -2;            // VT_QUAD_ROWS: -1 = rows packed back to back, 0..15 = force the row stride S of the bank-aware placement, -2 = planner
    int quad_grid2d = 1;           // VT_QUAD_GRID2D=0: 1-D grid with XCD-contiguous ids over all chunks (round-2 A/B)
    int quad_reverse = -1;         // VT_QUAD_REVERSE: 1 / 0 = the 2-D grid walks the in-plane tiles in descending / ascending order, -1 = planner
    bool test_fail_copy = false;   // VT_TEST_FAIL_COPY: pretend the secondary resident copies cannot be allocated (the re-planning path)
    int quad_perm = 1;             // VT_QUAD_PERM=0: identity lane -> pixel mapping in the plane-quad kernel (round-3 A/B)
    bool no_proj_cache = false;    // VT_NO_PROJ_CACHE: every projection recomputes the weighted plane sum (test of the cache)
    int quad_pingpong = 1;         // VT_QUAD_PINGPONG: 1 = every other launch of a handle walks the chunk layers from the last to the first; 0 / 2 = never / always
    int quad_zid = 1;              // VT_QUAD_ZID=0: trilinear launches with fz == 0 keep the two-plane kernel (round-3 A/B)
    int reorient = 4;              // VT_REORIENT: general matrices sample the resident copy whose rows follow the output's w axis, built at the n-th request (0 = never)
    int rows_pd = 8;               // VT_ROWS_PD=4: the row kernel's pixel tile is 4 x 8 (four waves) instead of 8 x 8
    int no_fused_relayout = 0;     // VT_NO_FUSED_RELAYOUT=1: the plane-quad forms of the in-plane transposed orientation through an exchanged plain copy (rounds 2-4) instead of relayout_zquad_swap12
    int rows_db = 0;               // VT_ROWS_DB: 0 = one run per workgroup (the default: faster), 1 = the row kernel walks a tile's runs with two row buffers where they fit, 2 = the same on the 4 x 8 tile
    int rows = 1;                  // VT_ROWS=0: maps that leave axis 2 alone take the axis-exchange path instead of the row kernel (kind 10); 2: the row kernel also for cubic launches with a fractional axis-2 offset (slower than the exchange path: tests only)
    int quad_zfir = 1;             // VT_QUAD_ZFIR=0: cubic launches with fz == 0 keep the four-plane kernel (round-4 A/B: the z-convolved copy)
    int zid_dch = 0;               // VT_ZID_DCH: chunk depth of the integer-offset trilinear kernel (0 = the trilinear default)
    int quad_nt = -1;              // VT_QUAD_NT: 1 / 0 = nontemporal / plain output stores of the plane-quad kernel, -1 = planner's choice
    bool no_block = false;         // VT_NO_BLOCK_KERNEL: general matrices on the round-1 box / packed kernels
    int block_rs = -1;             // VT_BLOCK_RS: force the row-stride index of the lane-block kernel (if it holds the box)
    bool block_linear = false;     // VT_BLOCK_LINEAR: trilinear general matrices on the lane-block kernel too (slower than packed footprints)
    bool block_no_trim = false;    // VT_BLOCK_NO_TRIM: the lane-block kernel stages whole boxes (A/B of the footprint trimming)
    int block_min = 240;           // VT_BLOCK_MIN: smallest output (cube edge) that general cubic launches take to the lane-block kernel
    int block_pad = -1;            // VT_BLOCK_PAD: plane-stride padding in floats instead of the bank model's choice
    double max_resident_gb = 0.0;  // VT_MAX_RESIDENT_GB: default resident-memory budget of every handle in GiB (0 = none); see vt_volume_set_max_resident
    int span = 1;                  // VT_SPAN=0: trilinear general matrices on round 1's packed-footprint kernel (vt_kernels_packed.hip) instead of round 5's (A/B)
    int span_pipe = 0;             // VT_SPAN_PIPE: 1 / 0 = trilinear general matrices on the software-pipelined / the single-buffer form of the packed-span kernel, -1 = the planner's cost model
    int block_th = 8;              // VT_BLOCK_TH: tile height of the lane-block kernel: 8 (8 x 8 x 16 tiles, four workgroups per CU; boxes beyond 40 KiB fall back to 16) or 16 (8 x 16 x 16, two per CU)
    void read()
    {
        auto num = [](const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; };
        tile = num("VT_TILE", -1);
#ifdef VT_LEGACY              // knobs of round 1's marching kernels: test build only
        la = num("VT_LA", 0);
        march_box = num("VT_MARCH_BOX", -1);
        if (march_box > 1) march_box = 1;
        lxpad = num("VT_LXPAD", -1);
#endif
        dch = std::max(0, num("VT_DCH", 0));
        blk_h = num("VT_BLK_H", -1);
        blk_w = num("VT_BLK_W", -1);
        plain_tile_order = num("VT_TILE_ORDER", 1) == 0;
        rswap_wfast = num("VT_RSWAP_WFAST", -1);
#ifdef VT_EXPERIMENTS         // ablation switches: `make EXTRA=-DVT_EXPERIMENTS` only
        exp_nostore = std::getenv("VT_EXP_NOSTORE") != nullptr;
        exp_noload = std::getenv("VT_EXP_NOLOAD") != nullptr;
        exp_nolds = std::getenv("VT_EXP_NOLDS") != nullptr;
        exp_noloop = std::getenv("VT_EXP_NOLOOP") != nullptr;
        exp_static = std::getenv("VT_EXP_STATIC") != nullptr;
        exp_notiles = std::getenv("VT_EXP_NOTILES") != nullptr;
        exp_stamps = std::getenv("VT_EXP_STAMPS") != nullptr;
#endif
        quad_nt = num("VT_QUAD_NT", -1);
#ifdef VT_LEGACY              // settled A/Bs of round 3 and the allocation-failure hook: test build only (the product build keeps the defaults)
        test_fail_copy = std::getenv("VT_TEST_FAIL_COPY") != nullptr;
        quad_perm = num("VT_QUAD_PERM", 1);
        quad_zid = num("VT_QUAD_ZID", 1);
        quad_grid2d = num("VT_QUAD_GRID2D", 1);
#endif
        quad_pingpong = num("VT_QUAD_PINGPONG", 1);
        quad_zfir = num("VT_QUAD_ZFIR", 1);
        rows = num("VT_ROWS", 1);
        rows_pd = num("VT_ROWS_PD", 8);
        rows_db = num("VT_ROWS_DB", 0);
        no_fused_relayout = num("VT_NO_FUSED_RELAYOUT", 0);
        reorient = num("VT_REORIENT", 4);
        no_proj_cache = std::getenv("VT_NO_PROJ_CACHE") != nullptr;
        zid_dch = num("VT_ZID_DCH", 0);
        quad_reverse = num("VT_QUAD_REVERSE", -1);
        quad_rows = num("VT_QUAD_ROWS", -2);
        no_block = std::getenv("VT_NO_BLOCK_KERNEL") != nullptr;
        block_rs = num("VT_BLOCK_RS", -1);
        block_linear = std::getenv("VT_BLOCK_LINEAR") != nullptr;
        block_pad = num("VT_BLOCK_PAD", -1);
        block_th = num("VT_BLOCK_TH", 8) == 16 ? 16 : 8;
        span_pipe = num("VT_SPAN_PIPE", 0);
        span = num("VT_SPAN", 1);
        { const char* e = std::getenv("VT_MAX_RESIDENT_GB"); max_resident_gb = e ? std::atof(e) : 0.0; }
        block_min = std::max(1, num("VT_BLOCK_MIN", 240));
        block_no_trim = std::getenv("VT_BLOCK_NO_TRIM") != nullptr;
    }
};


struct vt_volume {
    unsigned launch_no = 0;            // plane-quad launches so far (VT_QUAD_PINGPONG)
    int dev = 0;
    int interp = 0;
    int D = 0, H = 0, W = 0;           // resident source dims
    int oD = 0, oH = 0, oW = 0;        // output dims
    int64_t plane0 = 0;                // global index of resident plane 0
    int64_t gD = 0;                    // global depth
    int64_t out_plane0 = 0;            // global index of output plane 0
    int P = 0;                         // row pitch of d_src in floats: W rounded up to 4, pad columns hold 0
    float* d_src = nullptr;
    size_t src_bytes = 0;              // size of the d_src allocation (small ones are recycled per device)
    float* d_zeros = nullptr;          // 16 bytes of zeros: the border fetch target of the tiled kernel
    float* d_src_t = nullptr;          // resident copy with axes 0 and 1 exchanged (rotations about axis 1 march along it); lazy
    float* d_src_r = nullptr;          // resident copy transposed in-plane ([z][x][y], pitch Pr; quarter-turn class of in-plane maps); lazy
    int Pr = 0;
    float* d_src_x = nullptr;          // resident copy with axes 0 and 2 exchanged ([x][y][z], pitch Px; rotations about axis 2); lazy
    int Px = 0;
    float* d_tmp_x = nullptr;          // exchanged result of an axis-2 launch, before it is turned back
    size_t tmp_x_elems = 0;
    float* d_src_q = nullptr;          // plane-quad copies ([z/4][y][x][4]; vt_kernels_quad.hip) of the four orientations; lazy
    float* d_src_t_q = nullptr;
    float* d_src_r_q = nullptr;
    float* d_src_x_q = nullptr;
    size_t quad_bytes[4] = {0, 0, 0, 0};   // allocation sizes of the four quad copies (vt_volume_info)
    float* d_src_xe = nullptr;         // plain-layout copy convolved along axis 2 with the cubic weights of fraction 0 (row kernel, kind 10, cubic; vt_kernels_rows.hip: relayout_xfir); lazy
    float* d_src_qe[4] = {nullptr, nullptr, nullptr, nullptr};   // plane-quad copies of the Z-CONVOLVED volume per orientation (cubic launches with an integer axis-0 offset; vt_kernels_quad.hip: relayout_zquad_fir); lazy
    size_t quade_bytes[4] = {0, 0, 0, 0};
    int reorient_asked[3] = {0, 0, 0};   // general matrices that asked for the copy whose fastest axis is source axis 0 / 1 (vt_api.hip: try_general_reorient)
    int xe_retry_in = 0;               // calls to go before the x-convolved copy is attempted again after an allocation failure
    int copy_retry_in[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // calls to go before a quad copy that could not be allocated is attempted again (per orientation; 4..7: the z-convolved ones)
#ifdef VT_LEGACY
    // plane-pair copies of the four orientations (round 1's cubic marching kernel, kind 5): test build only
    float* d_src_zp = nullptr;
    float* d_src_t_zp = nullptr;
    float* d_src_r_zp = nullptr;
    float* d_src_x_zp = nullptr;
    int P2 = 0;                        // floats per pair-row of d_src_zp
#endif
    // lazily built resident copies: budget, least-recently-used eviction, build time (vt_api.hip: lazy copies)
    uint64_t max_resident = 0;         // bytes this handle may keep resident, plain copy included (0 = no limit); vt_volume_set_max_resident / VT_MAX_RESIDENT_GB
    uint64_t use_clock = 0;            // launches so far
    uint64_t copy_used[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // launch that last read copy i (LazyCopyId)
    float copies_ms = 0.f;             // GPU time spent building lazy copies so far
    int copies_built = 0, copies_evicted = 0;
    hipEvent_t evc0 = nullptr, evc1 = nullptr;
    float* spare = nullptr;            // the buffer of the lazy copy evicted last, kept for the next build of that size (a sweep under a budget
    size_t spare_bytes = 0;            // trades one orientation's copy for another's: no hipFree + hipMalloc of gigabytes per switch)
    int* d_queue = nullptr;            // lane-block kernel: tile counters (one per XCD + a departure count), zero between launches
    float* d_scratch_out = nullptr;    // staging for host outputs
    double* d_batch_m = nullptr;       // batch launches: n x 12 folded matrices
    std::vector<double> h_batch_m;     // ... and their host staging (must outlive the asynchronous upload)
    size_t batch_m_cap = 0;
    float* d_proj_tmp = nullptr;       // projection of general matrices: the transformed volume before the sum
    size_t proj_tmp_elems = 0;
    vt_volume* proj = nullptr;         // projection helper: 3 x H x W volume [S, S, S] sharing this handle's stream
    bool proj_sum_valid = false;       // the helper holds the weighted plane sum of (proj_sum_m3, proj_sum_oD, proj_sum_oplane0)
    double proj_sum_m3 = 0.0;
    int proj_sum_oD = 0;
    int64_t proj_sum_oplane0 = 0;
    int edge_pad = 0;                  // VT_EDGE_SCIPY: the resident copy carries this many mirrored voxels on every side (0 = texture contract)
    bool owns_stream = true;
    bool deferred = false;             // created with VT_SRC_DEFERRED: planes still being uploaded, not usable before vt_volume_finalize
    bool lo_interior = false;          // ... and its VT_SLAB_LO_INTERIOR flag, kept for the prefilter at finalize
    size_t scratch_elems = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float prefilter_ms = 0.f;
    int lds_limit = 160 * 1024;
    int cu_count = 256;
    // last launch, for vt_volume_info
    int last_kernel = 0, last_tile[3] = {0, 0, 0}, last_lds[3] = {0, 0, 0}, last_lds_bytes = 0, last_grid = 0;
    Tuning tune;                       // experiment overrides (environment, read at create)
};

namespace vt {

inline bool is_cubic(int interp) { return interp != VT_LINEAR; }
inline bool is_filtered(int interp) { return interp == VT_FILT_BSPLINE || interp == VT_FILT_BSPLINE_SIMPLE; }

// Row pitch of a resident plain-layout copy, in floats: the row's samples, at least one all-zero 16-byte vector after them
// (the border fetch target), rounded up so that EVERY ROW STARTS ON A 128-BYTE CACHE LINE.  With the round-1 pitch
// (roundup4(W) + 4: 2064 bytes at W = 512) every row segment a strided prefilter pass stores straddled cache lines:
// partial-line writes, [measured] 2.0 instead of 5.2 TB/s for tile-shaped stores (tools/probes/pattern_probe.hip, pitch 520)
// and 0.32 ms per strided pass at 512^3 whatever the load width.  Costs 6 % more resident bytes at 512, 3 % at 1024.
inline int resident_pitch(int W) { return (W + 4 + 31) & ~31; }

// The launch planner (vt_plan.hip): kernel family, tile shape, LDS budget and grid for one matrix on one handle.  Fills every
// field of `p` the chosen kernel reads; plan->kind = 1 (direct gather) when nothing tiled fits.  Host side, ~10 us.
void plan_launch(const vt_volume* v, const double m[12], int flags, AffineParams* p, TilePlan* plan);
bool plan_rows(const vt_volume* v, const double m[12], int flags, AffineParams* p, TilePlan* plan);

}  // namespace vt
