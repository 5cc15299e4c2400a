// vt_plan.hip -- the launch planner: which kernel family serves a matrix on a handle, with which tile, LDS budget and grid.
//
// One `plan_*` function per kernel family, tried in order of preference by `plan_launch` (kFamilies); each checks its own
// eligibility (matrix class, diagnostic flags, address ranges, LDS fit), fills the AffineParams fields its kernel reads and
// returns true, or leaves the plan untouched and returns false.  Host side, a few hundred flops (~10 us); the environment knobs
// of the experiments are gathered in vt_volume::tune (vt_host.h).
//
//   family   kind  kernel (file)                                   serves
//   rows       7   affine_rows         (vt_kernels_rows.hip)       maps that leave axis 2 alone, integer offset (plan_rows: planned by do_affine ahead of the rest)
//   quad       8   affine_march4       (vt_kernels_quad.hip)       axis-0-separable matrices, every interpolation
//   block      9   affine_block        (vt_kernels_block.hip)      general matrices, cubic, outputs >= 256^3
//   box        2   affine_tiled        (vt_kernels_affine.hip)     any matrix whose tile footprint's bounding box fits LDS
//   packed     6   affine_tiled_packed (vt_kernels_packed.hip)     invertible general matrices, trilinear (or forced)
//   direct     1   affine_direct       (vt_kernels_affine.hip)     everything else (tiny volumes, huge footprints)
// Test build only (`make LEGACY=1`): zpair 5 / march 4 (vt_kernels_march.hip: round 1's plane-pair and plain marching kernels) and
// kind 3 (affine_tiled_zsep, the box kernel with in-plane partial reuse) -- round 3's planner census (tools/planner_census.py) found
// them chosen for nothing but 4x in-plane minification of cubic volumes, which the box kernel serves.
#include "vt_host.h"
#include "vt_device.h"

#include <cmath>
#include <cstring>

namespace vt {
namespace {

struct PlanCtx {
    const vt_volume* v;
    const double* m;           // folded 3 x 4 matrix (resident coordinates)
    int flags;
    AffineParams* p;
    TilePlan* plan;
    bool cubic;
    bool zsep;                 // [1 0 0 tz; 0 a b ty; 0 c d tx]
    int halo2;                 // extra taps of the cubic stencil on each side (0 / 2 in total)
};

#ifdef VT_LEGACY      // helpers of the round-1 marching families (test build only)
// Chunk count of a marching launch.  Its workgroups all do the same work, so the chip runs them in rounds of `resident`
// workgroups and a launch of 5.33 rounds takes almost as long as one of 6 ([measured] 512^3 cubic at 0 / 30 degrees:
// 8 chunks = 5.33 rounds 0.274 / 0.304 ms; 6 chunks = 4.0 rounds 0.261 / 0.296; 3 chunks = 2.0 rounds 0.249 / 0.297;
// 4 chunks = 2.67 rounds 0.260 / 0.314; round-2 measurement).  Among chunk counts from n0/4 to 2*n0 take the one with the
// least rounds x planes marched per chunk, a partial round charged at its fraction + 0.3; launches of many rounds (>= 16)
// or of less than one keep n0, and so do planes with more tiles than the chip keeps resident (there deeper chunks
// separate in-plane neighbours in time and their shared rows miss L2: 640^3 cubic 0.544 vs 0.498 ms).  Used by the pair
// kernel only ([measured] 256^3 -9..-12 %, 384^3 -6..-10 %, 512^3 -2..-8 %).  `extra` = source planes a chunk stages beyond
// its output planes.
int64_t round_aware_chunks(int64_t oD, int g, int64_t inplane, int64_t resident, int64_t n0, int extra, int min_dch, int64_t n_floor)
{
    if (resident <= 0 || inplane > resident || inplane * n0 >= 16 * resident) return n0;
    auto cost = [&](int64_t n) {
        int64_t dch = (oD + n - 1) / n;
        dch = (dch + g - 1) / g * g;
        const int64_t n_act = (oD + dch - 1) / dch;
        const double R = (double)(inplane * n_act) / (double)resident;
        if (R < 1.0) return 1e300;
        const double fl = std::floor(R), fr = R - fl;
        return (fl + (fr > 1e-9 ? std::min(1.0, fr + 0.3) : 0.0)) * (double)(dch + extra);
    };
    int64_t best = n0;
    double best_c = cost(n0);
    if (best_c >= 1e300) return n0;
    best_c *= 0.97;                                   // switch only for a predicted gain of 3 % or more
    const int64_t lo = std::max<int64_t>(std::max<int64_t>(1, n_floor), n0 / 4);
    const int64_t hi = std::min<int64_t>(2 * n0, std::max<int64_t>(1, oD / std::max(1, min_dch)));
    for (int64_t n = hi; n >= lo; --n) {
        const double c = cost(n);
        if (c < best_c) { best_c = c; best = n; }
    }
    return best;
}

// Expected LDS cycles of one half-wave `ds_read2_b32` of the gather (relative to conflict-free = 1.0) when lanes
// step by (a, b) in (row, column) through an LDS image with row stride Lx: for each of the two dwords, the number
// of distinct addresses on the busiest of the 32 banks (identical addresses broadcast).  Averaged over a few
// sub-voxel offsets.  Used to choose the row stride of the marching cubic kernels, which are LDS-bound.
double gather_conflict_factor(double a, double b, int Lx, int ndw = 2)
{
    double total = 0;
    int samples = 0;
    for (int oy = 0; oy < 2; ++oy)
        for (int ox = 0; ox < 2; ++ox) {
            const double y0 = 8.0 + 0.37 * oy + 40.0 * std::fabs(std::min(a, 0.0)), x0 = 8.0 + 0.41 * ox + 40.0 * std::fabs(std::min(b, 0.0));
            for (int dw = 0; dw < ndw; ++dw) {
                int count[32];
                int addrs[32][32];
                for (int i = 0; i < 32; ++i) count[i] = 0;
                for (int l = 0; l < 32; ++l) {
                    const int addr = (int)std::floor(y0 + a * l) * Lx + (int)std::floor(x0 + b * l) + dw;
                    const int bank = addr & 31;
                    bool seen = false;
                    for (int k = 0; k < count[bank]; ++k) seen = seen || (addrs[bank][k] == addr);
                    if (!seen) addrs[bank][count[bank]++] = addr;
                }
                int worst = 1;
                for (int i = 0; i < 32; ++i) worst = std::max(worst, count[i]);
                total += worst;
                ++samples;
            }
        }
    return total / samples;
}


// Upper estimate of the packed footprint of a TH x TW in-plane tile (16-byte vectors per source plane) under rows 1, 2 of
// the matrix, as the marching kernel packs it: per source row the tapped span, aligned to 16 bytes.  The tile's sub-voxel
// position varies from tile to tile, so a 4 x 4 grid of offsets is sampled and a margin added; a tile that still exceeds
// the slot falls back to a direct gather inside the kernel, so the estimate affects speed only.
int estimate_packed_vectors(const double m[12], int th, int tw, int halo, int rows_cap, int* rows_out)
{
    const double a1 = m[5], b1 = m[6], a2 = m[9], b2 = m[10];       // d(sy)/dj, d(sy)/dk, d(sx)/dj, d(sx)/dk
    const double ia1 = march_recip(a1), ib1 = march_recip(b1);
    double neg1 = 0, neg2 = 0;
    for (double e : {a1 * (th - 1), b1 * (tw - 1)}) if (e < 0) neg1 += e;
    for (double e : {a2 * (th - 1), b2 * (tw - 1)}) if (e < 0) neg2 += e;
    int worst = 0, worst_rows = 0;
    for (int oy = 0; oy < 3; ++oy)
        for (int ox = 0; ox < 3; ++ox) {
            // box-relative base exactly as the kernel forms it: lo = base + neg, o = floor(lo) - halo, b = base - o
            const double fy0 = 0.33 * oy + 0.013, fx0 = 0.33 * ox + 0.017;
            const double by = fy0 - neg1 + halo, bx = fx0 - neg2 + halo;   // (+ up to 3 for the 16-byte alignment of o2)
            int total = 0, rows = 0;
            for (int Y = 0; Y < rows_cap; ++Y) {       // rows_cap: the bounding box's rows (+1)
                int mn, mx;
                if (!march_row_span(a1, b1, a2, b2, ia1, ib1, by, bx, Y, th, tw, halo, &mn, &mx)) continue;
                total += ((mx - mn) >> 2) + 2;                   // +1 vector: unknown 16-byte phase of the span start
                rows = Y + 1;
            }
            worst = std::max(worst, total);
            worst_rows = std::max(worst_rows, rows);
        }
    *rows_out = worst_rows + 1;
    return worst + worst / 64 + 2;
}


#endif  // VT_LEGACY

// Upper estimate of the packed footprint of a TH x TW in-plane tile in POSITIONS (plane-quad layout: one 16-byte vector per
// position, no alignment), as affine_march4 packs it.  A 3 x 3 grid of sub-voxel offsets is sampled; a row's span changes by
// at most one position with the offset, hence the margin of one position per row.  A tile that still exceeds the slot
// takes the kernel's direct-gather path (slow, never wrong).
int estimate_span_positions(const double m[12], int th, int tw, int halo, int rows_cap, int* rows_out)
{
    const double a1 = m[5], b1 = m[6], a2 = m[9], b2 = m[10];
    const double ia1 = march_recip(a1), ib1 = march_recip(b1);
    double neg1 = 0, neg2 = 0;
    for (double e : {a1 * (th - 1), b1 * (tw - 1)}) if (e < 0) neg1 += e;
    for (double e : {a2 * (th - 1), b2 * (tw - 1)}) if (e < 0) neg2 += e;
    int worst = 0, worst_rows = 0;
    for (int oy = 0; oy < 3; ++oy)
        for (int ox = 0; ox < 3; ++ox) {
            const double by = 0.33 * oy + 0.013 - neg1 + halo, bx = 0.33 * ox + 0.017 - neg2 + halo;
            int total = 0, rows = 0;
            for (int Y = 0; Y < rows_cap; ++Y) {
                int mn, mx;
                if (!march_row_span(a1, b1, a2, b2, ia1, ib1, by, bx, Y, th, tw, halo, &mn, &mx)) continue;
                total += mx - mn + 1;
                rows = Y + 1;
            }
            worst = std::max(worst, total);
            worst_rows = std::max(worst_rows, rows);
        }
    *rows_out = worst_rows + 1;
    return worst + worst_rows + 8;
}


// Row stride S (mod 16) for the cubic plane-quad kernel's bank-aware row starts.  A wave64 ds_read_b128 is served in four
// 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32: MI355X_MICROARCH.md, LDS), each conflict-free when its lanes
// hit distinct 16-byte slots mod 16 (equal addresses broadcast).  With rows placed at slot = column + row * S the slot of a lane's
// tap is (ix + iy * S) & 15 for its tap origin (iy, ix) -- the same for every tap of the 4 x 4 stencil.  The model evaluates two
// waves of a 256-thread workgroup at two sub-voxel tile positions and returns the S with the fewest LDS cycles per read
// (*factor: cycles relative to conflict-free).  ~3 us.
int quad_row_stride(const double m[12], int th, int tw, double* factor, bool lane_perm)
{
    // lane -> pixel position of the kernel (quad_lane_to_pos, vt_kernels_quad.hip): each service group = 16 consecutive positions
    auto lane_pos = [&](int lane) {
        if (!lane_perm) return lane;
        static const int sh[8] = {0, 3, 3, -2, 2, -3, -3, 0};
        return lane + 4 * sh[(lane >> 2) & 7];
    };
    static const int kGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const double a1 = m[5], b1 = m[6], a2 = m[9], b2 = m[10];
    double neg1 = 0, neg2 = 0;
    for (double e : {a1 * (th - 1), b1 * (tw - 1)}) if (e < 0) neg1 += e;
    for (double e : {a2 * (th - 1), b2 * (tw - 1)}) if (e < 0) neg2 += e;
    int iy[4][64], ix[4][64];                     // [sample][lane]
    int ns = 0;
    for (int off = 0; off < 2; ++off)
        for (int wave = 0; wave < 4; wave += 2, ++ns) {
            const double by = 0.29 + 0.42 * off - neg1 + 1.0, bx = 0.17 + 0.55 * off - neg2 + 1.0;
            for (int l = 0; l < 64; ++l) {
                const int pos = lane_pos(l), k = pos % tw, j = wave * (64 / tw) + pos / tw;
                iy[ns][l] = (int)std::floor(by + a1 * j + b1 * k);
                ix[ns][l] = (int)std::floor(bx + a2 * j + b2 * k);
            }
        }
    int best_s = 0;
    double best = 1e300;
    for (int S = 0; S < 16; ++S) {
        int cycles = 0;
        for (int smp = 0; smp < ns; ++smp)
            for (int half = 0; half < 2; ++half)
                for (int g = 0; g < 2; ++g) {
                    int cnt[16] = {0};
                    int seen_y[16][16], seen_x[16][16];
                    int worst = 1;
                    for (int i = 0; i < 16; ++i) {
                        const int l = kGroup[g][i] + 32 * half;
                        const int y = iy[smp][l], x = ix[smp][l];
                        const int slot = (x + y * S) & 15;
                        bool dup = false;
                        for (int q = 0; q < cnt[slot]; ++q) dup = dup || (seen_y[slot][q] == y && seen_x[slot][q] == x);
                        if (!dup) { seen_y[slot][cnt[slot]] = y; seen_x[slot][cnt[slot]] = x; ++cnt[slot]; }
                        worst = std::max(worst, cnt[slot]);
                    }
                    cycles += worst;
                }
        const double f = (double)cycles / (double)(ns * 4);
        if (f < best - 1e-9) { best = f; best_s = S; }
    }
    *factor = best;
    return best_s;
}

// Source extent of an in-plane TH x TW tile under rows 1, 2 of the matrix (+ taps), in rows / columns; false when absurd.
bool inplane_box(const double m[12], int th, int tw, int halo2, int L[3])
{
    const int T[3] = {1, th, tw};
    L[0] = 0;
    for (int r = 1; r < 3; ++r) {
        double ext = 0;
        for (int k = 1; k < 3; ++k) ext += std::fabs(m[4 * r + k]) * (T[k] - 1);
        if (!(ext < 4096.0)) return false;
        L[r] = (int)std::floor(ext) + 3 + halo2;
    }
    return true;
}

// neg / pos: lowest / highest source coordinate of a tile relative to its base, over tile axes k0 .. 2
void set_tile_reach(AffineParams* p, const double m[12], const int T[3], int k0)
{
    for (int r = 0; r < 3; ++r) {
        double neg = 0, pos = 0;
        for (int k = k0; k < 3; ++k) {
            const double e = m[4 * r + k] * (T[k] - 1);
            if (e < 0) neg += e; else pos += e;
        }
        p->neg[r] = neg; p->pos[r] = pos;
    }
}

void set_axis0_split(AffineParams* p, const double m[12])
{
    const double fl = std::floor(m[3]);
    p->zoff = (int32_t)fl;
    p->fz = (float)(m[3] - fl);
}

int experiment_flags(const vt_volume* v)
{
    return (v->tune.exp_nostore ? (1 << 21) : 0) | (v->tune.exp_noload ? (1 << 22) : 0) | (v->tune.exp_nolds ? (1 << 26) : 0) |
           (v->tune.exp_noloop ? (1 << 27) : 0) | (v->tune.exp_static ? (1 << 28) : 0) | (v->tune.exp_stamps ? (1 << 29) : 0) | (v->tune.exp_notiles ? (1 << 30) : 0);
}

// ---------------------------------------------------------------------------------------------------
// quad: marching kernel on the plane-quad layout (kind 8)
// ---------------------------------------------------------------------------------------------------
bool quad_pick_tile(PlanCtx& c, int64_t max_stride)
{
    const vt_volume* v = c.v;
    AffineParams* p = c.p;
    const int halo = c.cubic ? 1 : 0;
    // configurations in order of preference (first fit wins).  Cubic launches on planes beyond 512^2 prefer the 32 x 32 tile with
    // 512 threads (two pixels per thread as in the 16 x 32 tile, 8 % less halo per voxel): [measured, one process] 1024^3 sweep
    // 1.629 -> 1.603 ms; at 512^3 it wins at the quarter turns only (0.189 vs 0.201 ms) and loses 1-3 % elsewhere.
    // The integer-offset trilinear kernel prefers the same tile on planes up to 512^2 (large outputs only: half as many workgroups):
    // [measured, one variant per process, profiles/r03_process_ab.txt] 512^3 sweep 0.1924 -> 0.1876 ms on average (each form is bimodal
    // by 2 % from process to process), 1024^3 1.471 -> 1.482 (stays on 16 x 32); the cubic kernel at 512^3 loses 3 % with it.
    int order[8], norder = 0;
    // (Round 5: not the one-tap-plane cubic kernel, KIND 4 -- [measured, one process per variant, three alternations, three boxes] 1024^3
    //  filt_bspline sweep 1.508 / 1.509 / 1.510 ms on the 16 x 32 tile against 1.538 / 1.524 / 1.524 on 32 x 32 x 512 (1.3 %, the same sign in
    //  nine of nine pairs), 768^3 equal: its 30 KB of LDS keep four workgroups per CU.  The four-plane kernel keeps the preference.)
    const bool zfir_k4 = c.cubic && (float)(c.m[3] - std::floor(c.m[3])) == 0.0f && v->tune.quad_zfir != 0 && !(c.flags & VT_NO_ZFIR);
    const bool big_cubic = c.cubic && !zfir_k4 && (int64_t)v->H * v->W > 512LL * 512;
    const bool lin_zid = !c.cubic && (float)(c.m[3] - std::floor(c.m[3])) == 0.0f && v->tune.quad_zid != 0;
    const bool mid_linear = lin_zid && (int64_t)v->H * v->W <= 512LL * 512 && (int64_t)v->oD * v->oH * v->oW >= 384LL * 384 * 384;
    // ... and on larger planes the 16 x 64 tile with 512 threads (256-byte store rows): 1024^3 sweep 1.551 -> 1.529 ms (32 x 32: 1.482 vs
    // 1.471 for 16 x 32 on another box, i.e. worse there)
    const bool big_linear = lin_zid && (int64_t)v->H * v->W > 512LL * 512 && (int64_t)v->oD * v->oH * v->oW >= 384LL * 384 * 384;
    const int first = (big_cubic || mid_linear) ? 4 : (big_linear ? 5 : -1);
    if (first >= 0) order[norder++] = first;
    for (int cfg = 0; cfg < quad_config_count() && norder < 8; ++cfg)
        if (cfg != first) order[norder++] = cfg;
    for (int oi = 0; oi < norder; ++oi) {
        const int cfg = order[oi];
        if (v->tune.tile >= 0 && cfg != v->tune.tile) continue;
        int th, tw, nt;
        quad_config(cfg, &th, &tw, &nt);
        if ((int64_t)th * max_stride * 4 >= 0x7fffffffLL) continue;
        int L[3];
        if (!inplane_box(c.m, th, tw, c.halo2, L) || L[1] > march_rows_max()) continue;
        int rows = 0;
        const int npos = estimate_span_positions(c.m, th, tw, halo, L[1] + 1, &rows);
        if (rows > march_rows_max()) continue;
        int nvec64 = (npos + 63) & ~63;
        if (nvec64 > nt * quad_max_it()) continue;
        // cubic: bank-aware row starts (gaps of < 16 vectors per row, 7.5 on average) where the padded image still leaves four
        // workgroups per CU (the kernel's register allocation admits no more) -- i.e. up to ~25 degrees for a 16 x 32 tile; a tile
        // whose padded image overflows the slot packs its rows back to back instead (in the kernel)
        p->row_s = -1;
        // (a map without axis-1 shear along a pixel row, m[1][2] == 0, keeps every 16-pixel group inside one source row: conflict-free
        // however the rows are packed -- [measured] 0 degrees: 0.204 ms packed, 0.216 aligned)
        if (c.cubic && v->tune.quad_rows != -1 && (int64_t)v->oD * v->oH * v->oW >= 128LL * 128 * 128 &&
            (c.m[6] != 0.0 || v->tune.quad_rows >= 0 || v->tune.quad_perm == 0)) {
            const int padded64 = (npos + 9 * rows + 63) & ~63;
            if (padded64 <= nt * quad_max_it() && 2LL * padded64 * 16 <= (nt >= 512 ? 80 : 40) * 1024) {      // 4 (2) workgroups of 256 (512) threads per CU
                double f = 0;
                // With lanes assigned by service groups a group's taps are 16 consecutive pixels: column-aligned rows (S = 0: slot =
                // column mod 16) leave only the row crossings with an unchanged column as conflicts.  [measured, tools/r3_quad_ab.sh rows -> profiles/r03_ab_3_rows.txt, 512^3
                // cubic, 61 angles] S = 0: 0.2142 ms; the 16-candidate model's S: 0.2234; identity lanes + model (round 2): 0.2222;
                // rows packed back to back: 0.2242 (with 0.29-0.30 ms outliers at 39 / 51 degrees).  The model serves VT_QUAD_PERM=0.
                const int S = (v->tune.quad_rows >= 0) ? (v->tune.quad_rows & 15)
                              : (v->tune.quad_perm != 0 ? 0 : quad_row_stride(c.m, th, tw, &f, false));
                p->row_s = S;
                nvec64 = padded64;
            }
        }
        const int slot_bytes = nvec64 * 16;                   // two ring slots; slot 1 sits at offset slot_bytes (toggled with XOR)
        const int64_t bytes = std::max<int64_t>(2LL * slot_bytes, march_table_bytes());
        if (bytes > v->lds_limit) continue;
        c.plan->kind = 8; c.plan->cfg = cfg; c.plan->td = 4; c.plan->th = th; c.plan->tw = tw;
        c.plan->lds_bytes = (int)bytes;
        p->Lz = 2; p->Ly = std::min(L[1], march_rows_max()); p->Lx = npos; p->Lx_used = npos;
        p->slot_floats = slot_bytes / 4;
        return true;                              // configurations are listed in order of preference: first fit wins
    }
    return false;
}

bool plan_quad(PlanCtx& c)
{
    const vt_volume* v = c.v;
    AffineParams* p = c.p;
    TilePlan* plan = c.plan;
    if (!c.zsep || (c.flags & (VT_NO_MARCH | VT_NO_QUAD | VT_NO_ZPAIR))) return false;
    if (v->H > 65535 || (v->D + 3) / 4 > 65535) return false;      // relayout_zquad's grid (y = rows, z = quads): never plan what cannot be built
    // cubic with an integer axis-0 offset (every rotation about axis 0, every in-plane map): the four tap planes' weights are the constants
    // (1/6, 2/3, 1/6, 0), so the launch samples the z-convolved plane-quad copy with ONE tap plane per output plane (KIND 4)
    const bool zfir = c.cubic && (float)(c.m[3] - std::floor(c.m[3])) == 0.0f && v->tune.quad_zfir != 0 && !(c.flags & VT_NO_ZFIR);
    const int halo = (c.cubic && !zfir) ? 1 : 0;                  // axis-0 halo of the launch
    const int Wq = (v->W + 1 + 7) & ~7;                    // positions per quad-row: >= one zero position, rows of whole 128-byte lines
    const int64_t quad_bytes = (int64_t)v->H * Wq * 16;
    if (quad_bytes >= 0x7fffffffLL) return false;
    // output addressing of the kernel: 31-bit byte offsets inside a tile (rows) and inside a chunk (planes), for either
    // orientation of the output strides (axis-exchanged launches swap them afterwards)
    const int64_t max_stride = (int64_t)std::max(v->oD, v->oH) * v->oW;
    const TilePlan saved = *plan;
    if (!quad_pick_tile(c, max_stride)) return false;
    const int T[3] = {1, plan->th, plan->tw};
    set_tile_reach(p, c.m, T, 1);
    set_axis0_split(p, c.m);
    p->nTh = (v->oH + plan->th - 1) / plan->th;
    p->nTw = (v->oW + plan->tw - 1) / plan->tw;
    p->sPq = 4 * Wq;
    p->zero_off_q = v->W * 16;
    p->flags = (c.flags & VT_KEEP_OUTSIDE) | experiment_flags(v);
    if (v->tune.quad_nt < 0 ? true : v->tune.quad_nt != 0) p->flags |= (1 << 28);     // streaming output stores ([measured] +2..5 % in-process)
    if (v->tune.quad_perm != 0) p->flags |= (1 << 23);                                // lanes <-> pixels by ds_read_b128 service groups
    // trilinear with an integer axis-0 offset (every rotation about axis 0): one tap plane per output plane, no history quad
    const bool zid = !c.cubic && p->fz == 0.0f && v->tune.quad_zid != 0;
    if (zid) p->flags |= (1 << 25);
    if (zfir && p->fz == 0.0f) {
        p->flags |= (1 << 19);
        if (v->interp == VT_BSPLINE_SIMPLE || v->interp == VT_FILT_BSPLINE_SIMPLE) p->flags |= (1 << 18);
    }
    const bool zfir_on = (p->flags & (1 << 19)) != 0;
    const bool one_plane = zid || zfir_on;
    const int64_t inplane = (int64_t)p->nTh * p->nTw;
    // chunk depth: every chunk pays one quad step beyond its own planes (history of the first outputs), so chunks are deeper than
    // the plain kernels' -- but short-lived workgroups keep the write stream compact (tools/probes/pattern_probe.hip).
    // [measured, tools/march_ab.py] trilinear: 24 planes where a whole layer of tiles is resident at once (512^3: 0.204 vs 0.213 ms
    // at 76), 64 on larger planes (1024^3: 1.70 at 64, 1.77 at 32, 1.82 at 128); cubic, chunk starts aligned to quads (dshift):
    // 512^3 0.227 ms at 64 planes, 0.233 at 128, 0.236 at 256
    // (2-D grid, blockIdx.y = chunk: trilinear 1024^3 1.593 ms at 32 planes, 1.609 at 48, 1.663 at 64, 1.702 at 128; 512^3 flat 16..64)
    int target_dch = c.cubic ? 64 : (((int64_t)v->H * v->W <= 512 * 512) ? 24 : 32);
    // integer-offset trilinear: no history quad, so short chunks cost only their set-up; [measured, tools/r3_quad_ab.sh zid -> profiles/r03_ab_1_zid.txt] 1024^3: 1.476 ms at
    // 16 planes, 1.509 at 24, 1.518 at 32, 1.568 at 48, 1.595 at 64 (the copy structure alone behaves the same: front_probe);
    // 512^3 (variants in one process): 0.1961 at 24, 0.1979 at 16, 0.1992 at 32 -- within the handle-to-handle spread; decided per process below
    // (KIND 4, cubic on the z-convolved copy, keeps the cubic depth: its set-up -- the row-span table -- is the expensive one.  512^3 filt_bspline, one
    // process per variant: 0.2495 ms at 16 planes, 0.2047 at 32, 0.1975 at 64, 0.2103 at 128; the four-plane kernel 0.2046)
    // Round 5 re-check on the final kernel (tile origin in scalar registers, cheaper set-up), one process per variant, 60 angles, three alternations
    // (profiles/r05_headline_chunk_depth.txt): 512^3 0.2064 ms at 16 planes, 0.1905 at 24, 0.1873 at 32, 0.1869 at 40, 0.1880 at 48, 0.1923 at 64 --
    // 32 planes at every size now (1024^3 filt_bspline: 1.563 ms at 64 planes, 1.527 at 32; four-plane kernel: 1.594)
    if (zfir_on) target_dch = 32;
    if (zid) target_dch = 16;             // 512^3, one process per variant (profiles/r03_process_ab.txt): 0.1896 ms at 16, 0.1911 at 20, 0.1922 at 24, 0.1953 at 32
    if (one_plane && v->tune.zid_dch > 0) target_dch = v->tune.zid_dch;
    if (v->tune.dch > 0) target_dch = std::max(4, v->tune.dch);
    int64_t nchunks = std::max<int64_t>(1, (v->oD + target_dch - 1) / target_dch);
    // small volumes: shorter chunks until the launch has ~4 workgroups per CU, not below 8 planes per chunk
    if (v->tune.dch <= 0)
        nchunks = std::max(nchunks, std::min<int64_t>((4 * (int64_t)v->cu_count + inplane - 1) / inplane, (v->oD + 7) / 8));
    // scalar byte offsets: source quads of a chunk from its first quad, output planes from its first plane (31 bits each)
    const int64_t n_addr = std::max(((int64_t)(v->oD / 4 + 4) * quad_bytes) / 0x60000000LL + 1, ((int64_t)v->oD * max_stride * 4) / 0x60000000LL + 1);
    nchunks = std::max<int64_t>(nchunks, n_addr);
    plan->blocks_per_cu = quad_blocks_per_cu(plan->cfg, v->interp, plan->lds_bytes, one_plane);
    int dch = (int)((v->oD + nchunks - 1) / nchunks);
    dch = (dch + 3) & ~3;
    // chunk boundaries at c*dch + dshift: the first tap plane of every chunk but the first, d_begin + zoff - halo, is then the
    // first plane of a quad -- a cubic chunk marches dch/4 + 1 quads instead of dch/4 + 2 (64 planes: 17 steps instead of 18)
    const int dshift = (int)((((int64_t)halo - (int64_t)p->zoff) % 4 + 4) % 4);
    nchunks = (v->oD > dshift) ? (v->oD - dshift + dch - 1) / dch : 1;
    const int64_t grid = inplane * nchunks;
    if ((int64_t)(dch + dshift) * max_stride * 4 >= 0x7fffffffLL || grid > 0x7fffffffLL) { *plan = saved; return false; }
    p->dch = dch;
    p->dshift = dshift;
    p->nTd = (int)nchunks;
    if (v->tune.blk_h > 0 && v->tune.blk_w > 0) { p->blk_h = v->tune.blk_h; p->blk_w = v->tune.blk_w; }
    plan->grid = (int)grid;
    // 2-D grid (chunk in blockIdx.y): the plain tile orders only, reciprocals exact for every tile id of a layer
    const bool grid2d = (v->tune.quad_grid2d != 0) && p->blk_h <= 0 && nchunks <= 65535 && inplane * std::max(p->nTh, p->nTw) < (1LL << 32) &&
                        p->nTw > 1 && p->nTh > 1;
    if (grid2d) {
        p->nTw_magic = (uint32_t)((1ULL << 32) / (uint64_t)p->nTw + 1);
        p->nTh_magic = (uint32_t)((1ULL << 32) / (uint64_t)p->nTh + 1);
        p->flags |= (1 << 29);
        // descending tile order when the source rows run backwards as the tile row advances (m[1][1] < 0 on the launch's matrix)
        const bool rev = v->tune.quad_reverse >= 0 ? v->tune.quad_reverse != 0 : c.m[5] < 0.0;
        if (rev) p->flags |= (1 << 30);
    }
    return true;
}

#ifdef VT_LEGACY      // round-1 families: test build only (see kFamilies)
// ---------------------------------------------------------------------------------------------------
// zpair: cubic marching kernel on the plane-pair layout (kind 5)
// ---------------------------------------------------------------------------------------------------
bool zpair_pick_tile(PlanCtx& c)
{
    const vt_volume* v = c.v;
    AffineParams* p = c.p;
    const double* m = c.m;
    for (int cfg = 0; cfg < zpair_config_count(); ++cfg) {
        if (v->tune.tile >= 0 && cfg != v->tune.tile) continue;
        int th, tw, la, nt;
        zpair_config(cfg, &th, &tw, &la, &nt);
        if (v->tune.la > 0) la = std::min(3, v->tune.la);
        const int vec_max = nt * march_max_it();
        int L[3];
        if (!inplane_box(m, th, tw, c.halo2, L)) continue;
        L[2] = (L[2] + 1 + 1) & ~1;                      // origin aligned down by up to 1 position, even width
        // boxes (stride padding fetched from the zero vector) vs packed spans [measured]: 512^3 sweep mean 0.292 vs
        // 0.312 ms, 1024^3 2.25-2.33 vs 2.28-2.38 ms -> boxes; VT_MARCH_BOX=0 selects packed spans
        bool zp_box = true;
        if (v->tune.march_box >= 0) zp_box = v->tune.march_box != 0;
        const int lx_used = L[2];                         // columns that hold data; the rest is stride padding
        int best_lx = L[2];
        double best_f = 1e300;
        for (int pad = 0; pad <= 30; pad += 2) {         // bank-pair index = (y*Lx + x) mod 32 for ds_read_b64
            // [measured at 36 and 144 degrees] every 2 positions of padding cost ~3 % (LDS footprint), a
            // pathological stride costs 30-50 %: the model flags the pathological ones reliably
            if (pad > 0 && L[1] * ((L[2] + pad) / 2) > vec_max) break;     // the padded box must still be stageable
            const double f = gather_conflict_factor(m[6], m[10], L[2] + pad, 1) * (1.0 + 0.015 * pad);
            if (f < best_f - 1e-9) { best_f = f; best_lx = L[2] + pad; }
        }
        if (v->tune.lxpad >= 0) best_lx = L[2] + v->tune.lxpad;
        L[2] = best_lx;
        int slot_floats = L[1] * L[2] * 2;
        if (!zp_box) {
            int rows = 0;
            // packed spans: vectors of 2 positions; the estimate counts 4-position vectors, so double it (loose)
            const int vecs = 2 * estimate_packed_vectors(m, th, tw, 1, L[1] + 1, &rows);
            if (rows > march_rows_max() || L[1] > march_rows_max()) continue;
            slot_floats = vecs * 4;
            if (vecs > vec_max) continue;
        } else if (L[1] * (L[2] / 2) > vec_max) continue;
        // planes with far more tiles than the chip keeps resident: one more pair in flight where three slots still
        // leave three workgroups per CU ([measured] 1024^3: 2.020 -> 1.972 ms at 0 degrees, 2.221 -> 2.202 at 30;
        // 512^3, whole layers resident: 0.258 -> 0.263, so not there)
        if (v->tune.la <= 0 && la == 1 && 9LL * slot_floats * 4 <= 160 * 1024 &&
            (int64_t)((v->oH + th - 1) / th) * ((v->oW + tw - 1) / tw) > 7LL * v->cu_count)    // 640^3, 768^3: -2..-4 % with it
            la = 2;
        const int64_t bytes = std::max<int64_t>((int64_t)(la + 1) * slot_floats * 4, zp_box ? 0 : march_table_bytes());
        if (bytes > v->lds_limit) continue;
        c.plan->kind = 5; c.plan->cfg = cfg; c.plan->td = 2; c.plan->th = th; c.plan->tw = tw;
        c.plan->lds_bytes = (int)bytes;
        p->Lz = la + 1; p->Ly = L[1]; p->Lx = L[2]; p->Lx_used = lx_used;
        p->slot_floats = slot_floats;
        p->flags = (c.flags & VT_KEEP_OUTSIDE) | (zp_box ? (1 << 20) : 0) | experiment_flags(v);
        return true;
    }
    return false;
}

bool plan_zpair(PlanCtx& c)
{
    const vt_volume* v = c.v;
    AffineParams* p = c.p;
    TilePlan* plan = c.plan;
    if (!c.zsep || !c.cubic || (c.flags & (VT_NO_MARCH | VT_NO_ZPAIR))) return false;
    if (v->H > 65535 || (v->D + 1) / 2 > 65535) return false;      // relayout_zpair's grid
    if ((int64_t)v->H * v->P * 4 >= 0x7fffffffLL || (int64_t)v->H * (2 * (((v->W + 3) & ~3) + 4)) * 4 >= 0x7fffffffLL) return false;
    const TilePlan saved = *plan;
    if (!zpair_pick_tile(c)) return false;
    const int T[3] = {1, plan->th, plan->tw};
    set_tile_reach(p, c.m, T, 1);
    set_axis0_split(p, c.m);
    p->nTh = (v->oH + plan->th - 1) / plan->th;
    p->nTw = (v->oW + plan->tw - 1) / plan->tw;
    p->sP2 = 2 * (((v->W + 3) & ~3) + 4);
    p->zero_off2 = 2 * ((v->W + 3) & ~3) * 4;
    const int64_t inplane = (int64_t)p->nTh * p->nTw;
    int target_dch = ((int64_t)v->H * v->W <= 512 * 512) ? 64 : 32;
    if (v->tune.dch > 0) target_dch = std::max(2, v->tune.dch);
    int64_t nchunks = std::max<int64_t>(1, (v->oD + target_dch - 1) / target_dch);
    // small volumes: shorter chunks until the launch has ~4 workgroups per CU (a 128^3 volume has only 32
    // in-plane tiles: 64-plane chunks would leave 3/4 of the chip idle), but not below 8 planes per chunk
    if (v->tune.dch <= 0)
        nchunks = std::max(nchunks, std::min<int64_t>((4 * (int64_t)v->cu_count + inplane - 1) / inplane, (v->oD + 7) / 8));
    const int64_t pair_bytes = (int64_t)v->H * p->sP2 * 4;
    const int64_t n_addr = ((int64_t)(v->oD / 2 + 4) * pair_bytes) / 0x60000000LL + 1;   // 31-bit scalar offsets
    nchunks = std::max<int64_t>(nchunks, n_addr);
    if (v->tune.dch <= 0)
        nchunks = round_aware_chunks(v->oD, 2, inplane, (int64_t)v->cu_count * march_blocks_per_cu(true, plan->cfg, v->interp, plan->lds_bytes),
                                     nchunks, 4, 8, n_addr);
    int dch = (int)((v->oD + nchunks - 1) / nchunks);
    dch = (dch + 1) & ~1;
    nchunks = (v->oD + dch - 1) / dch;
    p->dch = dch;
    p->nTd = (int)nchunks;
    if (v->tune.blk_h > 0 && v->tune.blk_w > 0) { p->blk_h = v->tune.blk_h; p->blk_w = v->tune.blk_w; }
    const int64_t grid = inplane * nchunks;
    if (grid > 0x7fffffffLL) { *plan = saved; return false; }
    plan->grid = (int)grid;
    return true;
}

// ---------------------------------------------------------------------------------------------------
// march: marching kernel on the plain layout (kind 4)
// ---------------------------------------------------------------------------------------------------
bool march_pick_tile(PlanCtx& c)
{
    const vt_volume* v = c.v;
    AffineParams* p = c.p;
    const double* m = c.m;
    // Footprint staging: the linear kernels stage the packed row spans of the rotated tile (least traffic, least
    // LDS).  The cubic kernels are LDS-read-bound and sensitive to bank conflicts, which the irregular row starts
    // of the packed image make worse (measured 0.48 vs 0.39 ms at 45 degrees), so they stage the bounding box with
    // a conflict-aware row stride.  VT_MARCH_BOX=0/1 overrides.
    bool march_box = c.cubic;
    if (v->tune.march_box >= 0) march_box = v->tune.march_box != 0;
    for (int cfg = 0; cfg < march_config_count(); ++cfg) {
        if (v->tune.tile >= 0 && cfg != v->tune.tile) continue;
        int th, tw, g, la, nt;
        march_config(cfg, &th, &tw, &g, &la, &nt);
        if (v->tune.la > 0) la = v->tune.la;
        const int vec_max = nt * march_max_it();
        int L[3];
        if (!inplane_box(m, th, tw, c.halo2, L) || L[1] > march_rows_max()) continue;
        int rows = 0;
        const int vecs = estimate_packed_vectors(m, th, tw, c.cubic ? 1 : 0, L[1] + 1, &rows);
        if (vecs > vec_max || rows > march_rows_max()) continue;
        int slot_floats = vecs * 4;
        int lx_used4 = 0;
        if (march_box) {
            // full bounding box with the row stride (in 16-byte steps) that predicts the fewest bank conflicts for
            // this matrix' lane step (m[1][2], m[2][2])
            L[2] = (L[2] + 3 + 3) & ~3;
            lx_used4 = L[2];
            int best_lx = L[2];
            double best_f = 1e300;
            for (int pad = 0; pad <= 28; pad += 4) {
                if (pad > 0 && L[1] * (L[2] + pad) / 4 > vec_max) break;
                const double f = gather_conflict_factor(m[6], m[10], L[2] + pad) * (1.0 + 0.015 * pad);
                if (f < best_f - 1e-9) { best_f = f; best_lx = L[2] + pad; }
            }
            L[2] = best_lx;
            slot_floats = L[1] * L[2];
            if (L[1] * L[2] / 4 > vec_max) continue;
        }
        const int ring = (la + 1) * g + c.halo2 + 1;
        const int64_t bytes = std::max<int64_t>((int64_t)ring * slot_floats * 4, march_box ? 1024 : march_table_bytes());
        if (bytes > v->lds_limit) continue;
        // measured on MI355X (512^3 and 1024^3, 0..45 degrees): resident workgroups per CU matter more than lookahead depth
        // inside one workgroup, and a deeper ring or a smaller tile never paid off: configurations are listed in order of
        // preference and the first that fits wins unless VT_TILE forces one (planning is on the per-call path: keep it cheap)
        c.plan->kind = 4; c.plan->cfg = cfg; c.plan->td = g; c.plan->th = th; c.plan->tw = tw;
        c.plan->lds_bytes = (int)bytes;
        p->Lz = ring; p->Ly = std::min(L[1], march_rows_max()); p->Lx = L[2]; p->Lx_used = lx_used4 ? lx_used4 : L[2];
        p->slot_floats = slot_floats;
        p->flags = (c.flags & VT_KEEP_OUTSIDE) | (march_box ? (1 << 20) : 0) | experiment_flags(v);
        return true;
    }
    return false;
}

bool plan_march(PlanCtx& c)
{
    const vt_volume* v = c.v;
    AffineParams* p = c.p;
    TilePlan* plan = c.plan;
    if (!c.zsep || (c.flags & VT_NO_MARCH) || (int64_t)v->H * v->P * 4 >= 0x7fffffffLL) return false;
    const TilePlan saved = *plan;
    if (!march_pick_tile(c)) return false;
    const int T[3] = {1, plan->th, plan->tw};
    set_tile_reach(p, c.m, T, 1);
    set_axis0_split(p, c.m);
    p->nTh = (v->oH + plan->th - 1) / plan->th;
    p->nTw = (v->oW + plan->tw - 1) / plan->tw;
    const int g = plan->td;
    const int64_t inplane = (int64_t)p->nTh * p->nTw;
    // short chunks keep the workgroups that share source rows (in-plane neighbours) at nearby planes, so the
    // overlap of their boxes is served by the XCD's L2 instead of the fabric (measured: 1024^3 linear
    // 3.4 ms at 342 planes per chunk, 2.1 ms at 16); the cubic kernels pay 5 planes of prologue per chunk
    // [measured, 0 and 45 degrees] linear (packed spans): 16 planes at both 512^3 and 1024^3; cubic (boxes): 64 planes
    // at 512^3 (0.379 vs 0.387 ms), 32 at 1024^3 (2.80 vs 2.88 ms)
    const int target_dch = c.cubic ? (((int64_t)v->H * v->W <= 512 * 512) ? 64 : 32) : 16;
    int64_t nchunks = std::max<int64_t>(1, (v->oD + target_dch - 1) / target_dch);
    // small volumes: shorter chunks until the launch has ~4 workgroups per CU, not below 4 planes per chunk
    nchunks = std::max(nchunks, std::min<int64_t>((4 * (int64_t)v->cu_count + inplane - 1) / inplane, (v->oD + 3) / 4));
    // the chunk's planes are addressed with a 31-bit scalar byte offset from its first plane
    const int64_t plane_bytes = (int64_t)v->H * v->P * 4;
    const int64_t n_addr = ((int64_t)v->oD * plane_bytes) / 0x60000000LL + 1;
    nchunks = std::max<int64_t>(nchunks, n_addr);
    // (no round-aware chunk count here: [measured] the linear kernel loses more L2 sharing with deeper chunks than
    // it gains from whole rounds -- 512^3 0.250 vs 0.220 ms, 640^3 0.442 vs 0.419)
    if (v->tune.dch > 0) nchunks = std::max<int64_t>(1, (v->oD + v->tune.dch - 1) / v->tune.dch);
    int dch = (int)((v->oD + nchunks - 1) / nchunks);
    dch = ((dch + g - 1) / g) * g;
    nchunks = (v->oD + dch - 1) / dch;
    p->dch = dch;
    p->nTd = (int)nchunks;
    if (v->tune.blk_h > 0 && v->tune.blk_w > 0) { p->blk_h = v->tune.blk_h; p->blk_w = v->tune.blk_w; }
    const int64_t grid = inplane * nchunks;
    if (grid > 0x7fffffffLL) { *plan = saved; return false; }
    plan->grid = (int)grid;
    return true;
}

#endif  // VT_LEGACY

// ---------------------------------------------------------------------------------------------------
// box: 3-D tiles, bounding box of the footprint staged (kinds 2, 3)
// ---------------------------------------------------------------------------------------------------
// Picks the tile with the fewest staged bytes per output voxel; *bpv = that figure (the packed family compares against it).
bool pick_box_tile(PlanCtx& c, double* bpv)
{
    const vt_volume* v = c.v;
    const double* m = c.m;
    double best_cost = 1e300;
    bool found = false;
    for (int cfg = 0; cfg < tile_config_count(); ++cfg) {
        if (v->tune.tile >= 0 && cfg != v->tune.tile) continue;
        int T[3];
        tile_config(cfg, &T[0], &T[1], &T[2]);
        int L[3];
        bool ok = true;
        for (int r = 0; r < 3 && ok; ++r) {
            double ext = 0;
            for (int k = 0; k < 3; ++k) ext += std::fabs(m[4 * r + k]) * (T[k] - 1);
            if (!(ext < 4096.0)) { ok = false; break; }
            L[r] = (int)std::floor(ext) + 3 + c.halo2;       // floor(hi)-floor(lo) <= floor(ext)+1, +1 upper tap, +1 slack
        }
        if (!ok) continue;
        if (c.zsep) L[0] = T[0] + 1 + c.halo2;               // exactly the planes d0+zoff-halo .. d0+TD+zoff+halo
        L[2] = (L[2] + 3 + 3) & ~3;                          // origin aligned down by up to 3, stride multiple of 4
        const int64_t bytes = (int64_t)L[0] * L[1] * L[2] * 4;
        if (bytes > v->lds_limit) continue;
        const int blocks_per_cu = (int)std::min<int64_t>(8, (160 * 1024) / bytes);
        const double vox = (double)T[0] * T[1] * T[2];
        // staged bytes per output voxel, penalised when fewer than 3 workgroups fit a CU (no overlap of
        // one workgroup's staging with another's gather)
        const double cost = (double)bytes / vox * (blocks_per_cu >= 3 ? 1.0 : (blocks_per_cu == 2 ? 1.25 : 2.0));
        if (cost < best_cost) {
            best_cost = cost;
            *bpv = (double)bytes / vox;
            c.plan->kind = c.zsep ? 3 : 2; c.plan->cfg = cfg; c.plan->td = T[0]; c.plan->th = T[1]; c.plan->tw = T[2];
            c.plan->lds_bytes = (int)bytes;
            c.p->Lz = L[0]; c.p->Ly = L[1]; c.p->Lx = L[2];
            found = true;
        }
    }
    return found;
}

bool finish_box(PlanCtx& c)
{
    const vt_volume* v = c.v;
    AffineParams* p = c.p;
    TilePlan* plan = c.plan;
    if (c.zsep) set_axis0_split(p, c.m);
    const int T[3] = {plan->td, plan->th, plan->tw};
    set_tile_reach(p, c.m, T, 0);
    p->nTd = (v->oD + T[0] - 1) / T[0];
    p->nTh = (v->oH + T[1] - 1) / T[1];
    p->nTw = (v->oW + T[2] - 1) / T[2];
    const int64_t grid = (int64_t)p->nTd * p->nTh * p->nTw;
    if (grid > 0x7fffffffLL) return false;
    plan->grid = (int)grid;
    return true;
}

// ---------------------------------------------------------------------------------------------------
// packed: 3-D tiles, packed row spans of the footprint staged, persistent workgroups (kind 6)
// ---------------------------------------------------------------------------------------------------
struct PackedChoice { TilePlan plan; AffineParams p; double bpv; bool found; };

bool invert3(const double m[12], double inv[9])
{
    const double A[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
    const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    double amax = 0;
    for (double a : A) amax = std::max(amax, std::fabs(a));
    if (!(std::fabs(det) > 1e-6 * amax * amax * amax && amax < 64.0)) return false;
    const double id = 1.0 / det;
    const double r[9] = {(A[4] * A[8] - A[5] * A[7]) * id, (A[2] * A[7] - A[1] * A[8]) * id, (A[1] * A[5] - A[2] * A[4]) * id,
                         (A[5] * A[6] - A[3] * A[8]) * id, (A[0] * A[8] - A[2] * A[6]) * id, (A[2] * A[3] - A[0] * A[5]) * id,
                         (A[3] * A[7] - A[4] * A[6]) * id, (A[1] * A[6] - A[0] * A[7]) * id, (A[0] * A[4] - A[1] * A[3]) * id};
    for (int i = 0; i < 9; ++i) inv[i] = r[i];
    return true;
}

void pick_packed_tile(PlanCtx& c, const double inv[9], PackedChoice* out)
{
    const vt_volume* v = c.v;
    const double* m = c.m;
    double best = 1e300;
    out->found = false;
    // trilinear: round 5's kernel (vt_kernels_span.hip: 16-byte table entries, whole waves stage, at most 128 VGPRs = four workgroups per
    // CU); cubic (planned only when forced): round 1's (vt_kernels_packed.hip)
    const bool span = !c.cubic && v->tune.span != 0;
    const int ncfg = span ? span_config_count() : packed_config_count();
    for (int cfg = 0; cfg < ncfg; ++cfg) {
        if (v->tune.tile >= 0 && cfg != v->tune.tile) continue;
        int T[3];
        if (span) span_config(cfg, &T[0], &T[1], &T[2]); else packed_config(cfg, &T[0], &T[1], &T[2]);
        // too few tiles to amortise the per-workgroup set-up (see `enough` in plan_packed): do not even plan it -- the span
        // summation below is the most expensive part of the host-side planning (~20 us)
        const int64_t tiles_c = (int64_t)((v->oD + T[0] - 1) / T[0]) * ((v->oH + T[1] - 1) / T[1]) * ((v->oW + T[2] - 1) / T[2]);
        if (!(c.flags & VT_FORCE_PACKED) && tiles_c < 6 * (int64_t)v->cu_count * (c.cubic ? 2 : 3)) continue;
        PackGeom g;
        int L[3];
        bool ok = true;
        double neg[3], pos[3];
        for (int r = 0; r < 3 && ok; ++r) {
            double ext = 0;
            neg[r] = pos[r] = 0;
            for (int k = 0; k < 3; ++k) {
                const double e = m[4 * r + k] * (T[k] - 1);
                ext += std::fabs(e);
                if (e < 0) neg[r] += e; else pos[r] += e;
            }
            if (!(ext < 1000.0)) { ok = false; break; }
            g.ext[r] = ext;
            // (span kernel: its box origin lies 4e-9 below the tile's lowest coordinate and its Q32.32 coordinates carry 2e-9 of rounding,
            //  vt_kernels_span.hip: fx64 -- an extent within 1e-8 of an integer needs the next row as well)
            L[r] = (int)std::floor(ext + (span ? 1.0e-8 : 0.0)) + 3 + c.halo2;
        }
        if (!ok) continue;
        L[2] = (L[2] + 3 + 3) & ~3;
        const int rows = L[0] * L[1];
        if (rows > (span ? span_rows_max() : packed_rows_max()) || L[2] > 4000 || L[1] > 1023 || L[0] > 511) continue;
        // the kernel stages a box inside the volume through ONE buffer descriptor based at the box origin: the 32-bit byte offset of the
        // box's last row must stay below the descriptor's 2^31 - 1 records (planes of 4096 x 4128 floats reach that at 32 box planes;
        // plan_block has the same bound); the output tile's planes are addressed the same way
        if ((int64_t)L[0] * v->H * v->P * 4 >= 0x7fffffffLL) continue;
        if (span && (int64_t)(T[0] + 1) * v->oH * v->oW * 4 >= 0x7fffffffLL) continue;
        // ... and its rim tiles recover a staging vector's (z, y, x) inside the box by dividing its offset by the source's plane and row
        // sizes: the box must not be larger than the volume in y or x (tiny volumes: the other families serve them)
        if (span && (L[1] > v->H || L[2] > v->P)) continue;
        for (int i = 0; i < 9; ++i) g.inv[i] = inv[i];
        for (int cc = 0; cc < 3; ++cc) g.cst[cc] = inv[3 * cc] * neg[0] + inv[3 * cc + 1] * neg[1] + inv[3 * cc + 2] * neg[2];
        g.T[0] = T[0]; g.T[1] = T[1]; g.T[2] = T[2];
        g.halo = c.cubic ? 1 : 0;
        g.Lxbox = L[2];
        g.Lybox = L[1];
        int nvec = 0;
        for (int row = 0; row < rows; ++row) {
            int mn, mx;
            if (packed_row_span(g, row / L[1], row % L[1], &mn, &mx)) nvec += ((mx - (mn & ~3)) >> 2) + 1;
        }
        int cap_vec = nvec + rows / 16 + 8;                       // margin for host/device rounding differences
        if (span) cap_vec = (cap_vec + 63) & ~63;                 // whole waves stage
        if (cap_vec > (span ? span_vectors_max() : packed_vectors_max())) continue;
        const bool pipe = span && span_config_pipelined(cfg);     // software-pipelined tiles: two footprint buffers
        if (span && v->tune.span_pipe >= 0 && pipe != (v->tune.span_pipe != 0)) continue;
        if (pipe && cap_vec > 2560) continue;                      // (its descriptor list is written by 8 x 320 threads)
        const int table_floats = span ? (pipe ? 4 * rows + 84 : 4 * rows + 8) : (2 * rows + 8 + 3) & ~3;   // (wave-specialised: + four parameter slots)
        const int64_t bytes = ((int64_t)table_floats + (int64_t)cap_vec * 4 * (pipe ? 2 : 1) + (pipe ? cap_vec : 0)) * 4;   // (wave-specialised: + the descriptor list)
        if (bytes > v->lds_limit) continue;
        const int blocks_per_cu = (int)std::min<int64_t>(c.cubic ? 2 : (span ? 4 : 3), (160 * 1024) / bytes);   // VGPR-limited occupancy
        const double vox = (double)T[0] * T[1] * T[2];
        double cost = (double)bytes / vox * (blocks_per_cu >= 3 ? 1.0 : (blocks_per_cu == 2 ? 1.2 : 2.0));
        // [measured, 512^3, 24 of the reference's random rotations, per-matrix timings: tools/span_probe.py] the launch time follows the resident
        // workgroups per CU first (4: 0.34-0.35 ms, 3: 0.37-0.43, 2: 0.46-0.50), then the tile: at equal occupancy 16 x 8 x 32 (two whole
        // 128-byte output lines per store instruction) takes 0.03 ms less than 16 x 16 x 16, and an 8-deep tile pays the per-tile work twice
        if (span) cost *= (blocks_per_cu >= 4 ? 0.90 : 1.0) * (T[0] >= 16 ? 1.0 : 1.15) * (T[2] >= 32 ? 0.93 : 1.0);
        if (cost < best) {
            best = cost;
            out->found = true;
            out->bpv = (double)(((int64_t)table_floats + (int64_t)cap_vec * 4) * 4) / vox;      // staged bytes per voxel (one buffer)
            out->plan.kind = 6; out->plan.cfg = cfg; out->plan.td = T[0]; out->plan.th = T[1]; out->plan.tw = T[2];
            out->plan.lds_bytes = (int)bytes;
            out->p.Lz = L[0]; out->p.Ly = L[1]; out->p.Lx = cap_vec * 4;
            out->p.slot_floats = table_floats;
            for (int r = 0; r < 3; ++r) { out->p.neg[r] = neg[r]; out->p.pos[r] = pos[r]; }
            out->plan.geo = g;
            out->plan.blocks_per_cu = blocks_per_cu;
        }
    }
}

// General matrices: packed 3-D footprints when the linear part is invertible.  Bounding boxes are cheaper to address (no row
// table), so the packed form must stage clearly less to win: measured cross-over at ~0.6x of the box bytes per voxel for
// trilinear (512^3).  Cubic: both kernels gather with 8-byte reads and the boxes win at every size ([measured]
// rotation (25,-40,70): 250^3 0.177 vs 0.198 ms, 384^3 0.563 vs 0.604, 512^3 1.295 vs 1.328), so the packed form is only planned
// for trilinear (or when forced).  `have_box`: a box plan is already in *c.plan with box_bpv staged bytes per voxel.
bool plan_packed(PlanCtx& c, bool have_box, double box_bpv)
{
    const vt_volume* v = c.v;
    if (c.zsep || (c.flags & VT_NO_PACKED) || (c.cubic && !(c.flags & VT_FORCE_PACKED))) return false;
    double inv[9];
    if (!invert3(c.m, inv)) return false;
    PackedChoice pk;
    pk.plan = *c.plan;
    pk.p = *c.p;
    pick_packed_tile(c, inv, &pk);
    if (!pk.found) return false;
    const bool forced = (c.flags & VT_FORCE_PACKED) != 0;
    // every persistent workgroup pays ~20 us to build its span table and staging descriptors: worth it only when it
    // then walks several tiles (measured: boxes win up to 250^3, on par at 320^3, packed 1.5x ahead at 512^3)
    const int64_t pk_tiles = (int64_t)((v->oD + pk.plan.td - 1) / pk.plan.td) * ((v->oH + pk.plan.th - 1) / pk.plan.th) *
                             ((v->oW + pk.plan.tw - 1) / pk.plan.tw);
    const bool enough = pk_tiles >= 6 * (int64_t)v->cu_count * std::max(1, pk.plan.blocks_per_cu);
    // A source that fits the 256 MB memory-side cache is re-read from there, and the box kernel's plainer gather wins unless the
    // footprint is much smaller than the box; a larger source makes the staged bytes HBM bytes, and the footprint wins whenever
    // it is smaller at all.  [measured, 100 random rotations, trilinear, ms where the old rule chose boxes: boxes / footprints]
    // 384^3 (226 MB) 0.211 / 0.220, 512^3 (537 MB) 0.517 / 0.475 (tools/general_tiles.py)
    const bool beyond_cache = (int64_t)v->D * v->H * v->P * 4 > (256LL << 20);
    const double margin = c.cubic ? 0.5 : (beyond_cache ? 1.0 : 0.6);
    if (!(!have_box || forced || (enough && pk.bpv < margin * box_bpv))) return false;
    *c.plan = pk.plan;
    *c.p = pk.p;
    AffineParams* p = c.p;
    p->nTd = (v->oD + pk.plan.td - 1) / pk.plan.td;
    p->nTh = (v->oH + pk.plan.th - 1) / pk.plan.th;
    p->nTw = (v->oW + pk.plan.tw - 1) / pk.plan.tw;
    const int64_t ntiles = (int64_t)p->nTd * p->nTh * p->nTw;
    if (ntiles > 0x7fffffffLL) { c.plan->kind = 1; return true; }      // (as before: such a launch goes to the direct kernel)
    // persistent workgroups: as many as stay resident, a multiple of 8 (one share per XCD)
    if (v->tune.plain_tile_order) p->flags |= (1 << 23);
    p->flags |= experiment_flags(v);
    int64_t nwg = std::min<int64_t>(ntiles, (int64_t)v->cu_count * std::max(1, c.plan->blocks_per_cu));
    nwg = std::max<int64_t>(8, (nwg + 7) / 8 * 8);
    c.plan->grid = (int)nwg;
    // One id per fetch: consecutive tiles are neighbours along w, which share source cache lines and complete each other's output
    // lines; staged by different workgroups at the same time they meet in the L2 ([measured, 512^3 trilinear, 100 random rotations]
    // 4 ids per fetch: L2 hit 0.22, 2.47 GB of HBM-side traffic; 1: 0.56, 1.54 GB; static striding: 0.35, 2.19 GB).  Making the
    // output axis that follows source x the fastest one instead of w loses 3 % (the output lines are completed later).
    p->dch = 1;
    // the kernel's tile decode divides super-block indices by the super-block counts along w and h: (u * magic) >> 32 == u / n
    p->nTw_magic = (uint32_t)((1ULL << 32) / (uint64_t)((p->nTw + 3) >> 2) + 1);
    p->nTh_magic = (uint32_t)((1ULL << 32) / (uint64_t)((p->nTh + 3) >> 2) + 1);
    return true;
}

// ---------------------------------------------------------------------------------------------------
// block: 8 x 8 x 16 tiles, bank-tuned LDS box, compact lane blocks, persistent workgroups (kind 9)
// ---------------------------------------------------------------------------------------------------
// LDS cycles of one `ds_read_b64` of the cubic gather (1.0 = conflict-free) for a 32-lane group of the lane map: the worst
// bank pair's count of distinct 8-byte words.  The 48 reads of a voxel are the same 32 addresses shifted by constants, so one
// read per tile position is the whole gather.  Linear interpolation reads 4-byte words from 32 banks.
double block_conflicts(const double m[12], int lm, int RS, int PS, bool cubic)
{
    static const double kBase[4][3] = {{8.13, 8.27, 8.41}, {8.44, 8.50, 8.78}, {8.71, 8.09, 8.33}, {8.92, 8.66, 8.05}};
    double tot = 0;
    for (int b = 0; b < 4; ++b) {
        int words[32], n = 0;
        for (int l = 0; l < 32; ++l) {
            const int t[3] = {lm == 0 ? (l >> 4) : 0, lm == 0 ? ((l >> 2) & 3) : (l >> 4), lm == 0 ? (l & 3) : (l & 15)};
            int f[3];
            for (int r = 0; r < 3; ++r)
                f[r] = (int)std::floor(kBase[b][r] + 24.0 + m[4 * r] * t[0] + m[4 * r + 1] * t[1] + m[4 * r + 2] * t[2]);
            const int a = f[0] * PS + f[1] * RS + (cubic ? (((f[2] - 1) & ~1) >> 1) * 2 : f[2]);
            const int word = cubic ? (a >> 1) : a;
            bool seen = false;
            for (int i = 0; i < n; ++i) seen = seen || words[i] == word;
            if (!seen) words[n++] = word;
        }
        int cnt[32] = {0}, worst = 0;
        for (int i = 0; i < n; ++i) worst = std::max(worst, ++cnt[((words[i] % 32) + 32) % 32]);
        tot += worst;
    }
    return tot / 4.0;
}

bool plan_block_th(PlanCtx& c, int th);

// the half-height tile first where it is asked for (its box must fit 40 KiB), else round 2's full-height tile
bool plan_block(PlanCtx& c)
{
    if (c.v->tune.block_th == 8 && plan_block_th(c, 8)) return true;
    return plan_block_th(c, 16);
}

bool plan_block_th(PlanCtx& c, int th)
{
    const vt_volume* v = c.v;
    const double* m = c.m;
    AffineParams* p = c.p;
    TilePlan* plan = c.plan;
    if ((c.flags & (VT_NO_BLOCK | VT_NO_PACKED | VT_FORCE_PACKED)) || v->tune.no_block) return false;
    if (c.zsep) return false;            // the axis-0-separable box kernel reuses in-plane partial sums: 4 / 16 LDS reads per voxel
    // trilinear: staging dominates (8 taps per voxel against 25 staged bytes), and the packed-footprint kernel stages a third of a
    // box: [measured] 512^3 general rotation 0.78 ms here, 0.52 ms packed.  VT_BLOCK_LINEAR=1 keeps the path testable.
    if (!c.cubic && !v->tune.block_linear) return false;
    // persistent workgroups want many tiles each: [measured, 100 random rotations, ms, lane blocks vs boxes] 160^3 0.071 / 0.050,
    // 200^3 0.112 / 0.090, 250^3 0.158 / 0.161, 288^3 0.212 / 0.223, 384^3 0.473 / 0.553, 512^3 1.05 / 1.23, 640^3 2.10 / 2.62
    // (tools/general_ab.py, VT_BLOCK_MIN).  Round 4, half-height tiles: 200^3 0.109 / 0.090, 250^3 0.150 / 0.161, 288^3 0.194 -> the threshold is 240^3
    if (!(c.flags & VT_FORCE_TILED) && (int64_t)v->oD * v->oH * v->oW < (int64_t)v->tune.block_min * v->tune.block_min * v->tune.block_min) return false;
    // tile height 8 (8 x 8 x 16 voxels, boxes of at most 40 KiB, four workgroups per CU) or 16 (round 2's 8 x 16 x 16, two per CU)
    int T[3];
    T[1] = th;
    block_tile(th, &T[0], &T[2]);
    int L[3];
    for (int r = 0; r < 3; ++r) {
        double ext = 0;
        for (int k = 0; k < 3; ++k) ext += std::fabs(m[4 * r + k]) * (T[k] - 1);
        if (!(ext < 256.0)) return false;
        L[r] = (int)std::floor(ext) + 3 + c.halo2;           // floor(hi)-floor(lo) <= floor(ext)+1, +1 upper tap, +1 slack
    }
    const int lx_used = (L[2] + 3 + 3) & ~3;                 // origin aligned down by up to 3, whole vectors
    // Row stride: the smallest of 28 / 36 floats that holds the box row -- or 32 where the box then exceeds the staging budget
    // (a stride of 32 floats puts rows two apart on the same banks: more conflicts, but 5 of the reference's 100 random rotations
    // otherwise fall back to the bounding-box kernel at 1.38 instead of ~1.1 ms).  VT_BLOCK_RS forces an index.
    int rs_idx = -1, RS = 0, PS = 0;
    int64_t vectors = 0;
    auto try_rs = [&](int idx) {
        if (idx < 0 || idx >= block_rs_count() || (th == 16 && idx > 2) || block_rs(idx) < lx_used) return false;
        const int rs = block_rs(idx);
        // plane stride: the padding (whole vectors, one bank period) with the fewest predicted gather conflicts
        int best_pad = 0;
        double best_f = 1e300;
        for (int pad = 0; pad < 64; pad += 4) {
            if ((int64_t)L[0] * (L[1] * rs + pad) / 4 > block_max_vectors(th)) break;       // (a padding the staging budget cannot hold)
            const double f = block_conflicts(m, 0, rs, L[1] * rs + pad, c.cubic) * (1.0 + 0.002 * pad);
            if (f < best_f - 1e-9) { best_f = f; best_pad = pad; }
        }
        if (v->tune.block_pad >= 0) best_pad = v->tune.block_pad & ~3;
        const int ps = L[1] * rs + best_pad;
        const int64_t vec = (int64_t)L[0] * ps / 4;
        if (vec > block_max_vectors(th)) return false;
        rs_idx = idx; RS = rs; PS = ps; vectors = vec;
        return true;
    };
    if (v->tune.block_rs >= 0) {
        if (!try_rs(v->tune.block_rs)) return false;
    } else {
        const int* order = nullptr;
        const int norder = block_rs_order(th, &order);                          // the tile height's order of preference
        bool ok = false;
        for (int i = 0; i < norder && !ok; ++i) ok = try_rs(order[i]);
        if (!ok) return false;
    }
    const int64_t plane_b = (int64_t)v->H * v->P * 4;
    if ((int64_t)L[0] * plane_b >= 0x7fffffffLL || (int64_t)T[0] * v->oH * v->oW * 4 >= 0x7fffffffLL) return false;
    const int box_bytes = (int)((vectors + 63) / 64 * 64 * 16);
    const int lds_bytes = box_bytes + 16;
    if (lds_bytes > v->lds_limit) return false;

    plan->kind = 9; plan->cfg = rs_idx; plan->td = T[0]; plan->th = T[1]; plan->tw = T[2];
    plan->lds_bytes = lds_bytes;
    p->Lz = L[0]; p->Ly = L[1]; p->Lx = RS; p->Lx_used = lx_used; p->Lps = PS;
    p->flags = (c.flags & VT_KEEP_OUTSIDE) | experiment_flags(v);
    set_tile_reach(p, m, T, 0);
    // footprint trimming (invertible linear part): the kernel stages, per box row, only the columns packed_row_span proves reachable
    double inv[9];
    if (!v->tune.block_no_trim && invert3(m, inv)) {
        PackGeom& g = plan->geo;
        for (int i = 0; i < 9; ++i) g.inv[i] = inv[i];
        for (int cc = 0; cc < 3; ++cc) g.cst[cc] = inv[3 * cc] * p->neg[0] + inv[3 * cc + 1] * p->neg[1] + inv[3 * cc + 2] * p->neg[2];
        for (int r = 0; r < 3; ++r) {
            g.ext[r] = p->pos[r] - p->neg[r];
            g.T[r] = T[r];
        }
        g.halo = c.cubic ? 1 : 0;
        g.Lxbox = lx_used;
        g.Lybox = L[1];
        p->flags |= (1 << 25);
    }
    p->nTd = (v->oD + T[0] - 1) / T[0];
    p->nTh = (v->oH + T[1] - 1) / T[1];
    p->nTw = (v->oW + T[2] - 1) / T[2];
    p->psv_magic = (uint32_t)(4294967296.0 / (double)(PS / 4)) + 1u;
    p->lds_cap = box_bytes;
    // the kernel's tile decode divides super-block indices by the super-block counts along w and h: (u * magic) >> 32 == u / n
    p->nTw_magic = (uint32_t)((1ULL << 32) / (uint64_t)((p->nTw + 3) >> 2) + 1);
    p->nTh_magic = (uint32_t)((1ULL << 32) / (uint64_t)((p->nTh + 3) >> 2) + 1);
    // steps between a thread's eight voxels (Gray order): +8 w, +8 h, -8 w, +4 d, -8 h; a backward step is the exact negative of
    // the forward one, so that the walk closes
    const int col[5] = {2, 1, 2, 0, 1};
    const double mul[5] = {8.0, 8.0, -8.0, 4.0, -8.0};
    const int neg_of[5] = {-1, -1, 0, -1, 1};
    for (int s = 0; s < 5; ++s)
        for (int r = 0; r < 3; ++r) {
            if (neg_of[s] >= 0) {
                const uint64_t fwd = ((uint64_t)(uint32_t)p->binc_hi[neg_of[s]][r] << 32) | p->binc_lo[neg_of[s]][r];
                const uint64_t back = 0 - fwd;
                p->binc_hi[s][r] = (int32_t)(uint32_t)(back >> 32); p->binc_lo[s][r] = (uint32_t)back;
                continue;
            }
            const double step = m[4 * r + col[s]] * mul[s];
            const double fl = std::floor(step);
            double lo = std::floor((step - fl) * 4294967296.0 + 0.5);
            int32_t hi = (int32_t)fl;
            if (lo >= 4294967296.0) { lo = 0; hi += 1; }
            p->binc_hi[s][r] = hi; p->binc_lo[s][r] = (uint32_t)lo;
        }
    plan->blocks_per_cu = (int)std::min<int64_t>(th == 16 ? 2 : 4, (160 * 1024) / lds_bytes);
    const int64_t ids = blocked_tile_count(p->nTd, p->nTh, p->nTw);
    int64_t grid = std::min<int64_t>((int64_t)v->cu_count * plan->blocks_per_cu, ids);
    grid = std::max<int64_t>(8, (grid + 7) / 8 * 8);              // persistent workgroups, the same number on every XCD
    // ids per queue fetch: 4 where every workgroup serves many tiles, 1 on small grids (a chunk of 4 would leave workgroups idle)
    p->dch = (int)std::max<int64_t>(1, std::min<int64_t>(4, ids / (grid * 8)));
    plan->grid = (int)grid;
    return true;
}

// box and packed compete on staged bytes per voxel
bool plan_general(PlanCtx& c)
{
    double box_bpv = 1e300;
    const bool have_box = pick_box_tile(c, &box_bpv);
    if (plan_packed(c, have_box, box_bpv)) return true;
    if (!have_box) return false;
    // Strong minification: a tile's bounding box grows with the cube of the scale while most of the output maps outside the volume.  Where
    // the box leaves one workgroup per CU, or (trilinear) stages more than ~43 bytes per voxel, the direct kernel's gather through the
    // caches is faster -- [measured, 512^3, tools/diag/magnify_ab.py -> profiles/r05_minification_routing.txt] trilinear scale 2.25 / 2.5 / 3:
    // boxes 0.403 / 0.516 / 0.919 ms against 0.372 / 0.369 / 0.352; cubic scale 2.5 / 3: 1.10 / 1.15 against 0.63 / 0.55; the one case
    // measured the other way is a cubic in-plane rotation at scale 2 (2.01 against 2.26 ms).  VT_FORCE_TILED keeps the boxes.
    // Only where the map shrinks the volume to a ninth or less (|det| >= 9: nearly all of the output lies outside, which is what makes the
    // direct kernel cheap): a mild minification with a rotation -- scale 1.2, rotation (10, 20, 30): a box of more than 80 KB, 58 % of the
    // output inside -- takes 6.3 ms on the direct kernel (cubic, 512^3) and stays with the boxes.
    const double* mm = c.m;
    const double det = mm[0] * (mm[5] * mm[10] - mm[6] * mm[9]) - mm[1] * (mm[4] * mm[10] - mm[6] * mm[8]) + mm[2] * (mm[4] * mm[9] - mm[5] * mm[8]);
    if (c.plan->kind == 2 && !(c.flags & VT_FORCE_TILED) && c.v->tune.tile < 0 && std::fabs(det) >= 9.0 &&
        (c.plan->lds_bytes > 80 * 1024 || (!c.cubic && box_bpv > 43.0))) {
        c.plan->kind = 1; c.plan->cfg = -1; c.plan->td = c.plan->th = c.plan->tw = 0; c.plan->lds_bytes = 0; c.plan->grid = 0;
        return true;
    }
    if (!finish_box(c)) { c.plan->kind = 1; }
    return true;
}

typedef bool (*plan_fn)(PlanCtx&);
struct Family { const char* name; plan_fn plan; };
const Family kFamilies[] = {
    {"quad", plan_quad},        // axis-0-separable, plane-quad layout
#ifdef VT_LEGACY
    {"zpair", plan_zpair},      // axis-0-separable cubic, plane-pair layout (test build)
    {"march", plan_march},      // axis-0-separable, plain layout (test build)
#endif
    {"block", plan_block},      // 3-D tiles: lane blocks on a bank-tuned box (rotations, mild scale / shear)
    {"general", plan_general},  // 3-D tiles: bounding boxes vs packed footprints
};

}  // namespace

// the fields every family reads: matrix, dims, output strides, the skirt rule's valid intervals
static void plan_prepare(const vt_volume* v, const double m[12], int flags, AffineParams* p, TilePlan* plan)
{
    std::memcpy(p->m, m, sizeof(double) * 12);
    p->sD = v->D; p->sH = v->H; p->sW = v->W; p->sP = v->P;
    for (int r = 0; r < 3; ++r) {
        // Q32.32 split of the depth-axis step m[r][0] (|m| < 4096 is checked per family for tiled launches)
        const double step = m[4 * r];
        const double fl = std::floor(step);
        p->inc_hi[r] = (std::fabs(step) < 2.0e9) ? (int32_t)fl : 0;
        p->inc_lo[r] = (uint32_t)std::min(4294967295.0, std::floor((step - fl) * 4294967296.0 + 0.5));
        if ((step - fl) * 4294967296.0 + 0.5 >= 4294967296.0) { p->inc_lo[r] = 0; p->inc_hi[r] += 1; }
    }
    p->oD = v->oD; p->oH = v->oH; p->oW = v->oW;
    p->ostride = (int64_t)v->oH * v->oW; p->orow = v->oW;
    p->ord[0] = 0; p->ord[1] = 1; p->ord[2] = 2;
    p->ia1 = march_recip(m[5]); p->ib1 = march_recip(m[6]);
    p->flags = (flags & VT_KEEP_OUTSIDE);
    if (v->edge_pad > 0) {
        // scipy's constant mode: a coordinate outside [0, dim-1] gives 0 (transforms.py:147-152); on the padded resident copy
        // that is [R, dim_p - R - 1], written as a half-open interval ending one ulp above the last valid coordinate
        const double R = (double)v->edge_pad;
        const int dims[3] = {v->D, v->H, v->W};
        for (int r = 0; r < 3; ++r) {
            p->vlo[r] = R;
            p->vhi[r] = std::nextafter((double)dims[r] - R - 1.0, 1.0e300);
        }
    } else {
        // skirt rule src + 0.5 in [0, dim) on the global volume, expressed on resident coordinates
        p->vlo[0] = -0.5 - (double)v->plane0;  p->vhi[0] = (double)v->gD - 0.5 - (double)v->plane0;
        p->vlo[1] = -0.5;                      p->vhi[1] = (double)v->H - 0.5;
        p->vlo[2] = -0.5;                      p->vhi[2] = (double)v->W - 0.5;
    }
    p->zero_off = ((v->W + 3) & ~3) * 4;

    plan->kind = 1; plan->cfg = -1; plan->td = plan->th = plan->tw = 0; plan->lds_bytes = 0; plan->grid = 0;
}

// ---------------------------------------------------------------------------------------------------
// rows: maps that leave axis 2 alone (kind 10, vt_kernels_rows.hip).  Planned by do_affine ahead of the axis exchanges, on the handle's
// own orientation only; false = not this class / does not fit, the plan is untouched (kind 0) and the usual dispatch follows.
// ---------------------------------------------------------------------------------------------------
bool plan_rows(const vt_volume* v, const double m[12], int flags, AffineParams* p, TilePlan* plan)
{
    const int64_t n_out = (int64_t)v->oD * v->oH * v->oW;
    if (flags & (VT_FORCE_DIRECT | VT_NO_ZSEP | VT_NO_MARCH | VT_FORCE_XSWAP | VT_FORCE_PACKED | VT_NO_ROWS)) return false;
    if (v->tune.rows == 0) return false;
    if (n_out < 64 * 64 * 64 && !(flags & VT_FORCE_TILED)) return false;
    // [a b 0 t0; c e 0 t1; 0 0 1 t], t an integer multiple of four (16-byte staging vectors); not also axis-0-separable (that class
    // has the plane-quad kernel: pure translations, the identity)
    if (!(m[2] == 0.0 && m[6] == 0.0 && m[8] == 0.0 && m[9] == 0.0 && m[10] == 1.0)) return false;
    if (m[0] == 1.0 && m[1] == 0.0 && m[4] == 0.0) return false;
    // any finite offset along axis 2 (round 5): a multiple of four starts a run on a 16-byte vector of the source row (16 vectors per
    // row), every other offset stages 18; a fractional one adds the x taps of the interpolation (vt_kernels_rows.hip: kinds 2 / 3)
    const double t = m[11];
    if (!(std::fabs(t) < 1.0e9)) return false;
    const double tfl = std::floor(t);
    const bool frac = t != tfl;
    const int nv = (!frac && ((int64_t)tfl & 3) == 0) ? 16 : 18;
    // ... except cubic with a FRACTIONAL offset, which stays with the exchange path unless asked for (VT_ROWS=2, the parity tests): its 64
    // taps per voxel from the plain copy are [measured, 512^3, one box, one process per case, tools/diag/rows_shift_ab.sh] 0.594 / 0.610 ms at
    // 33 / 80 degrees against 0.440 / 0.439 for exchanged copy + plane-quad kernel + transpose pass.  (Integer offsets: 0.328 / 0.303 at a
    // multiple of four, 0.368 / 0.327 at t = 2, against 0.424 / 0.421; trilinear 0.219 / 0.221, 0.260 / 0.243 at t = 2, 0.277 / 0.261 at t = 0.5,
    // against 0.32-0.34 for the bounding-box kernel.)
    if (frac && is_cubic(v->interp) && v->tune.rows != 2) return false;
    for (int i = 0; i < 12; ++i) if (!(std::fabs(m[i]) < 1.0e9)) return false;
    int ph, run;
    rows_tile(&ph, &run);
    // the pixel tile: 8 x 8 (eight waves) where its box fits, else 4 x 8 -- [measured, 512^3, 33 degrees, one box] trilinear 0.266 -> 0.239 ms,
    // cubic 0.369 -> 0.358 (a quarter less staged per voxel); VT_ROWS_PD=4 keeps the smaller one
    const bool cubic = is_cubic(v->interp);
    const int halo2 = cubic ? 2 : 0;
    int pd = 0, L[2] = {0, 0}, lds = 0;
    for (int cand = (v->tune.rows_pd == 4 ? 4 : 8); cand >= 4 && pd == 0; cand -= 4) {
        if ((v->oD + cand - 1) / cand > 65535 || (v->oH + ph - 1) / ph > 65535) continue;
        bool ok = true;
        for (int r = 0; r < 2 && ok; ++r) {
            const double ext = std::fabs(m[4 * r]) * (cand - 1) + std::fabs(m[4 * r + 1]) * (ph - 1);
            ok = ext < 200.0;
            L[r] = (int)std::floor(ext + 1.0e-8) + 3 + halo2;      // floor(hi) - floor(lo - 1e-9) <= floor(ext + 1e-8) + 1, + 1 upper tap, + 1 slack
        }
        lds = L[0] * L[1] * 4 * nv * 4 + cand * ph * 16 * 4;  // the staged rows + 16 dwords per pixel (one wave per d, eight pixels each)
        if (ok && lds <= 64 * 1024 && lds <= v->lds_limit) pd = cand;      // (strong minification in the (d, h) plane: the general kernels serve it)
    }
    if (pd == 0) return false;
    // Two row buffers (round 5, VT_ROWS_DB=1; 2 = on the 4 x 8 tile): a workgroup walks all runs of its pixel tile, staging run k + 1 while
    // it computes run k.  Built because the one-run form waits half of its wave cycles (SQ_WAIT_ANY 50 %), and SLOWER: [measured, 512^3,
    // one variant per process, tools/rows_ab.sh] trilinear 0.250 against 0.218 ms at 33 degrees (0.251 / 0.226 at 80), cubic 0.416 against
    // 0.331 (0.413 / 0.307); on 4 x 8 tiles 0.279 / 0.443.  Two boxes are 80-108 KiB: ONE workgroup per CU, and what hides a workgroup's
    // staging on this chip is another workgroup's compute, not its own next buffer (the same answer as for the lane-block kernel in round 3
    // and the packed-span kernel this round).  Kept behind the knob; the conditions: the two boxes fit, the output has at least three
    // runs, (oH / 8) x (oD / pd) workgroups fill the chip, a tile's eight output rows stay below the descriptor's 2^31 bytes.
    bool db = false;
    if (v->tune.rows_db != 0 && (v->oW + run - 1) / run >= 3 && (int64_t)ph * v->oW * 4 < 0x7fffffffLL) {
        auto box_lds = [&](int cand, int* Lz, int* Ly) {
            int LL[2];
            for (int r = 0; r < 2; ++r) {
                const double ext = std::fabs(m[4 * r]) * (cand - 1) + std::fabs(m[4 * r + 1]) * (ph - 1);
                LL[r] = (int)std::floor(ext + 1.0e-8) + 3 + halo2;
            }
            *Lz = LL[0]; *Ly = LL[1];
            return 2 * LL[0] * LL[1] * 4 * nv * 4 + cand * ph * 16 * 4;
        };
        int Lz2, Ly2;
        for (int cand = (v->tune.rows_db == 2 ? 4 : pd); cand >= 4 && !db; cand -= 4) {
            const int l2 = box_lds(cand, &Lz2, &Ly2);
            const int64_t wgs = (int64_t)((v->oH + ph - 1) / ph) * ((v->oD + cand - 1) / cand);
            if (l2 <= v->lds_limit && l2 <= 160 * 1024 && wgs >= 2 * (int64_t)v->cu_count && (v->oD + cand - 1) / cand <= 65535) {
                db = true; pd = cand; L[0] = Lz2; L[1] = Ly2; lds = l2;
            }
        }
    }
    plan_prepare(v, m, flags, p, plan);
    const int T[3] = {pd, ph, 1};
    set_tile_reach(p, m, T, 0);
    p->Lz = L[0]; p->Ly = L[1]; p->Lx = 4 * nv; p->Lx_used = run;
    p->psv_magic = (uint32_t)(4294967296.0 / (double)L[1]) + 1u;
    p->zoff = (int32_t)tfl;
    p->fz = 0.0f;
    p->flags = (flags & VT_KEEP_OUTSIDE) | (frac ? (1 << 16) : 0) | (db ? (1 << 17) : 0);
    if (v->interp == VT_BSPLINE_SIMPLE || v->interp == VT_FILT_BSPLINE_SIMPLE) p->flags |= (1 << 18);
    plan->kind = 10; plan->cfg = 0; plan->td = pd; plan->th = ph; plan->tw = run;
    plan->lds_bytes = lds;
    plan->grid = (int)std::min<int64_t>(0x7fffffff, (int64_t)(db ? 1 : (v->oW + run - 1) / run) * ((v->oH + ph - 1) / ph) * ((v->oD + pd - 1) / pd));
    plan->blocks_per_cu = std::max(1, std::min(8, (160 * 1024) / std::max(1, lds)));
    return true;
}

void plan_launch(const vt_volume* v, const double m[12], int flags, AffineParams* p, TilePlan* plan)
{
    const int64_t n_out = (int64_t)v->oD * v->oH * v->oW;
    plan_prepare(v, m, flags, p, plan);
    bool want_tiled = n_out >= 64 * 64 * 64;
    if (flags & VT_FORCE_TILED) want_tiled = true;
    if (flags & VT_FORCE_DIRECT) want_tiled = false;
    if (!want_tiled) return;

    PlanCtx c;
    c.v = v; c.m = m; c.flags = flags; c.p = p; c.plan = plan;
    c.cubic = is_cubic(v->interp);
    c.halo2 = c.cubic ? 2 : 0;           // cubic taps reach one voxel further on each side
    c.zsep = !(flags & VT_NO_ZSEP) && m[0] == 1.0 && m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[8] == 0.0 && std::fabs(m[3]) < 1.0e9;
    for (const Family& f : kFamilies) {
        if (f.plan(c)) return;
#ifndef VT_LEGACY
        // the round-1 axis-0-separable families (plain / plane-pair marching, the box kernel with in-plane partial reuse) exist in the
        // test build only: a separable matrix the plane-quad kernel declines is planned as a general matrix
        if (f.plan == plan_quad) c.zsep = false;
#endif
    }
    plan->kind = 1;                       // nothing tiled fits: direct gather
}

}  // namespace vt
