// vt_device.h -- device helpers shared by the transform kernels (gfx950).
#pragma once
#include "vt_internal.h"

namespace vt {

// ---------------------------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------------------------

// bspline.h:102-112
__host__ __device__ __forceinline__ void bspline_weights(float f, float& w0, float& w1, float& w2, float& w3)
{
    const float one_frac = 1.0f - f;
    const float squared = f * f;
    const float one_sqd = one_frac * one_frac;
    w0 = (1.0f / 6.0f) * one_sqd * one_frac;
    w1 = (2.0f / 3.0f) - 0.5f * squared * (2.0f - f);
    w2 = (2.0f / 3.0f) - 0.5f * one_sqd * (2.0f - one_frac);
    w3 = (1.0f / 6.0f) * squared * f;
}

// bspline.h:114-122, evaluated at the four tap offsets -1,0,1,2 of cubicTex3DSimple
// (helper_interpolation.h:51-61): t = |offset - f| lands in the [1,2), [0,1), (0,1], (1,2] branches.
__host__ __device__ __forceinline__ float bspline_fn(float t)
{
    t = fabsf(t);
    const float a = 2.0f - t;
    return (t < 1.0f) ? ((2.0f / 3.0f) - 0.5f * t * t * a) : ((t < 2.0f) ? (a * a * a * (1.0f / 6.0f)) : 0.0f);
}

template <bool SIMPLE>
__host__ __device__ __forceinline__ void cubic_weights(float f, float (&w)[4])
{
    if constexpr (SIMPLE) {
        w[0] = bspline_fn(-1.0f - f);
        w[1] = bspline_fn(0.0f - f);
        w[2] = bspline_fn(1.0f - f);
        w[3] = bspline_fn(2.0f - f);
    } else {
        bspline_weights(f, w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------------------------------
// tiled kernel
// ---------------------------------------------------------------------------------------------------

// Blocks b and b+8 share an XCD (round-robin dispatch).  Map the blocks of one XCD onto a contiguous
// range of tile ids (bijective for any grid size).  Placement only affects speed, never results.
// Wave64 inclusive scans on the DPP network: row_shr inside the 16-lane rows, then row_bcast:15 (rows 1, 3 take the last lane of rows 0, 2)
// and row_bcast:31 (rows 2, 3 take lane 31) -- six dependent vector operations where a __shfl_up ladder is six LDS round trips
// (ds_bpermute_b32).  Lanes without a source keep `old` (bound_ctrl off), the operation's identity.  All 64 lanes must be active.
__device__ __forceinline__ int wave_scan_add(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ int wave_scan_max(int x)      // values >= -1
{
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(-1, x, 0x143, 0xc, 0xf, false));
    return x;
}
// lane i takes lane i - 1's value, lane 0 takes `first` (wave_shr:1)
__device__ __forceinline__ int wave_shift_up1(int x, int first) { return __builtin_amdgcn_update_dpp(first, x, 0x138, 0xf, 0xf, false); }

__device__ __forceinline__ int xcd_contiguous(int b, int n)
{
    const int xcd = b & 7, q = n >> 3, r = n & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (b >> 3);
}


// Tile order of the marching kernels inside one chunk layer of nTh x nTw in-plane tiles:
//   * w fastest (default), or h fastest when the source is the in-plane transposed copy (flag bit 24): consecutive tiles
//     then read neighbouring source rows instead of rows TW apart;
//   * blocked (p.blk_h > 0): blocks of blk_h x blk_w tiles, w fastest inside a block, blocks running DOWN a block column
//     first.  The workgroups resident on one XCD are consecutive tiles; when a layer has more tiles than the chip keeps
//     resident, a few stacked blocks make their union a compact patch whose rotated footprints overlap inside the XCD's
//     L2 instead of a 1024-wide strip of tiles whose boxes share little (1024^3 cubic at 30 degrees read 2.4x the
//     algorithmic bytes from HBM with the strip order).  Partial blocks at the right / bottom edges are decoded exactly,
//     so the grid has no padding ids.
__device__ __forceinline__ void march_tile(const AffineParams& p, int t, int& th_i, int& tw_i, int& chunk)
{
    const int per_layer = p.nTh * p.nTw;
    chunk = t / per_layer;
    const int u = t - chunk * per_layer;
    if (p.blk_h > 0) {
        const int col_tiles = p.nTh * p.blk_w;            // tiles in a full block column
        const int bc = u / col_tiles;
        const int wc = min(p.blk_w, p.nTw - bc * p.blk_w);
        const int u1 = u - bc * col_tiles;
        const int blk_tiles = p.blk_h * wc;
        const int br = u1 / blk_tiles;
        const int u2 = u1 - br * blk_tiles;
        const int r = u2 / wc;
        th_i = br * p.blk_h + r;
        tw_i = bc * p.blk_w + (u2 - r * wc);
    } else if (p.flags & (1 << 24)) {
        tw_i = u / p.nTh;
        th_i = u - tw_i * p.nTh;
    } else {
        th_i = u / p.nTw;
        tw_i = u - th_i * p.nTw;
    }
}

// 3-D tiles in 4 x 4 x 4 super-blocks (w fastest inside a block and across blocks): consecutive tile ids -- the tiles
// that are in flight together on one XCD -- form a compact region of the volume, so the footprints that neighbouring
// tiles share (in all three directions) are fetched from HBM once and then hit the XCD's L2.  With the plain
// (d, h, w) order a tile's d-neighbour is nTh*nTw tiles away and its share of the footprint always misses.
// Returns false for ids that fall outside the tile grid (partial super-blocks); ids run to blocked_tile_count().
__host__ __device__ __forceinline__ int blocked_tile_count(int nTd, int nTh, int nTw)
{
    return ((nTd + 3) >> 2) * ((nTh + 3) >> 2) * ((nTw + 3) >> 2) * 64;
}
__device__ __forceinline__ bool blocked_tile(int t, int nTd, int nTh, int nTw, int& td, int& th, int& tw)
{
    const int nSh = (nTh + 3) >> 2, nSw = (nTw + 3) >> 2;
    const int sb = t >> 6, l = t & 63;
    const int sw = sb % nSw, s2 = sb / nSw;
    const int sh = s2 % nSh, sd = s2 / nSh;
    td = sd * 4 + (l >> 4);
    th = sh * 4 + ((l >> 2) & 3);
    tw = sw * 4 + (l & 3);
    return td < nTd && th < nTh && tw < nTw;
}

// Q32.32 fixed-point coordinate: hi = integer part (box index), lo = fraction.  Stepping along the tile's
// depth axis is two full-rate integer adds per axis instead of float64 arithmetic; the split into
// (index, fraction) is free.  For ordinary matrices (float32 entries of moderate magnitude) the arithmetic
// is exact; otherwise the drift is < 2^-29 voxel over a tile column.
struct Fx { int hi; unsigned lo; };

__device__ __forceinline__ Fx to_fx(double x)
{
    const double fl = floor(x);
    Fx r;
    r.hi = (int)fl;
    r.lo = (unsigned)((x - fl) * 4294967296.0);
    return r;
}

__device__ __forceinline__ void fx_step(Fx& c, int inc_hi, unsigned inc_lo)
{
    const unsigned lo = c.lo + inc_lo;
    c.hi += inc_hi + (lo < c.lo ? 1 : 0);
    c.lo = lo;
}

__device__ __forceinline__ float fx_frac(const Fx& c) { return (float)c.lo * 0x1p-32f; }


// ---------------------------------------------------------------------------------------------------
// cubic gather from an LDS image: 16 rows x 4 x-taps with 8-byte reads
// ---------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// The four x taps [ix-1, ix+2] of a row are fetched as three 8-byte-aligned ds_read_b64 (256 B/clk/CU) instead of four
// ds_read_b32 (128 B/clk/CU): with x1 = ix-1, par = x1 & 1, e = x1 - par, the taps sit at e+par .. e+par+3 of the six loaded
// values, and meet the weights shifted by the parity.  When par = 0 the third pair is not needed and re-reads the second
// one (weight 0), so nothing outside the row is touched.  `row(c, bb)` returns the LDS byte address of column e of tap
// row (z tap c, y tap bb); rows must start 8-byte aligned.  Inline asm: hipcc would fuse the adjacent 8-byte loads into
// ds_read2_b64 (half the bytes per LDS cycle).  Two z planes of taps are in flight: the wait for plane c leaves the 12
// reads of plane c+1 outstanding.  [measured: 384^3 cubic general rotation 0.75 -> 0.57 ms]
template <typename RowAddr>
__device__ __forceinline__ float cubic_gather_b64(RowAddr row, int par, const float (&wx)[4], const float (&wy)[4], const float (&wz)[4])
{
    const unsigned off3 = par ? 16u : 8u;                          // bytes: third pair, or the second one again
    v2f t[2][12];
    auto issue = [&](int c, v2f (&r)[12]) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const unsigned ra = row(c, bb);
            // (no wait states needed around this statement: its inputs are address VGPRs from integer VALU instructions, which the hardware
            //  interlocks; see the note at vt_kernels_block.hip: lds_read_b64)
            asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %4"
                         : "=&v"(r[3 * bb]), "=&v"(r[3 * bb + 1]), "=&v"(r[3 * bb + 2]) : "v"(ra), "v"(ra + off3));
        }
    };
    issue(0, t[0]);
    // The 16 tap rows are summed per COLUMN first (three pairs of column sums, packed FMAs on the pairs as read), the four x weights
    // meet the four columns of the stencil once at the end, picked by the parity with selects: the one or two extra columns of the
    // aligned window never enter the result (a zero weight would turn a non-finite neighbour into NaN: 0 * inf), and a voxel costs 48
    // packed FMAs instead of 80 scalar ones.
    v2f S0 = {0.f, 0.f}, S1 = {0.f, 0.f}, S2 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        v2f (&r)[12] = t[c & 1];
        if (c < 3) {
            issue(c + 1, t[(c + 1) & 1]);
            asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]),
                         "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]),
                         "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]));
        }
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float w = wz[c] * wy[bb];
            S0 = __builtin_elementwise_fma(r[3 * bb], (v2f)(w), S0);
            S1 = __builtin_elementwise_fma(r[3 * bb + 1], (v2f)(w), S1);
            S2 = __builtin_elementwise_fma(r[3 * bb + 2], (v2f)(w), S2);           // par = 0: the second pair again, never selected below
        }
    }
    const float t0 = par ? S0.y : S0.x, t1 = par ? S1.x : S0.y, t2 = par ? S1.y : S1.x, t3 = par ? S2.x : S1.y;
    return fmaf(wx[3], t3, fmaf(wx[2], t2, fmaf(wx[1], t1, wx[0] * t0)));
}

__device__ __forceinline__ unsigned lds_byte_address(const float* p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}

// ---------------------------------------------------------------------------------------------------
// the skirt rule, one definition for every kernel
// ---------------------------------------------------------------------------------------------------
// A voxel is inside when vlo <= src < vhi on every axis, with src evaluated by exactly this fma chain -- the one
// affine_direct and the oracle use.  The tiled kernels form their tap coordinates incrementally (tile base + offsets,
// fixed-point steps), which differs from this chain by an ulp; a voxel that lands exactly on the skirt (quarter turns about
// a half-integer centre do that) must not change sides with the kernel, so every per-voxel test goes through here and
// tiles are only classified as wholly inside / outside when they clear the skirt by kTileMargin.
constexpr double kTileMargin = 1.0e-6;

__device__ __forceinline__ double canonical_coord(const AffineParams& p, int r, int d, int h, int w)
{
    // p.ord: the launch's columns in the order of the ORIGINAL problem's axes (identity unless axes were exchanged)
    const int c0 = p.ord[0], c1 = p.ord[1], c2 = p.ord[2];
    const double x0 = (double)(c0 == 0 ? d : (c0 == 1 ? h : w));
    const double x1 = (double)(c1 == 0 ? d : (c1 == 1 ? h : w));
    const double x2 = (double)(c2 == 0 ? d : (c2 == 1 ? h : w));
    return fma(p.m[4 * r + c0], x0, fma(p.m[4 * r + c1], x1, fma(p.m[4 * r + c2], x2, p.m[4 * r + 3])));
}
__device__ __forceinline__ bool canonical_inside_axis(const AffineParams& p, int r, int d, int h, int w)
{
    const double s = canonical_coord(p, r, d, h, w);
    return (s >= p.vlo[r]) && (s < p.vhi[r]);
}
__device__ __forceinline__ bool canonical_inside(const AffineParams& p, int d, int h, int w)
{
    return canonical_inside_axis(p, 0, d, h, w) && canonical_inside_axis(p, 1, d, h, w) && canonical_inside_axis(p, 2, d, h, w);
}

// ---------------------------------------------------------------------------------------------------
// direct (untiled) sampling from global memory with explicit border tests
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fetch0(const float* __restrict__ src, const AffineParams& p, int z, int y, int x)
{
    if ((unsigned)z < (unsigned)p.sD && (unsigned)y < (unsigned)p.sH && (unsigned)x < (unsigned)p.sW)
        return src[((int64_t)z * p.sH + y) * p.sP + x];
    return 0.0f;
}


// One output value at integer tap origin (iz,iy,ix) and fractions (fz,fy,fx), taps fetched from global memory.
template <int KIND>
__device__ __forceinline__ float direct_sample(const float* __restrict__ src, const AffineParams& p,
                                               int iz, int iy, int ix, float fz, float fy, float fx)
{
    if constexpr (KIND == 0) {
        const float a000 = fetch0(src, p, iz, iy, ix), a001 = fetch0(src, p, iz, iy, ix + 1);
        const float a010 = fetch0(src, p, iz, iy + 1, ix), a011 = fetch0(src, p, iz, iy + 1, ix + 1);
        const float a100 = fetch0(src, p, iz + 1, iy, ix), a101 = fetch0(src, p, iz + 1, iy, ix + 1);
        const float a110 = fetch0(src, p, iz + 1, iy + 1, ix), a111 = fetch0(src, p, iz + 1, iy + 1, ix + 1);
        const float x00 = fmaf(fx, a001 - a000, a000);
        const float x01 = fmaf(fx, a011 - a010, a010);
        const float x10 = fmaf(fx, a101 - a100, a100);
        const float x11 = fmaf(fx, a111 - a110, a110);
        const float y0 = fmaf(fy, x01 - x00, x00);
        const float y1 = fmaf(fy, x11 - x10, x10);
        return fmaf(fz, y1 - y0, y0);
    } else {
        float wx[4], wy[4], wz[4];
        cubic_weights<KIND == 2>(fx, wx);
        cubic_weights<KIND == 2>(fy, wy);
        cubic_weights<KIND == 2>(fz, wz);
        float val = 0.f;
        for (int c = 0; c < 4; ++c) {
            float accy = 0.f;
            for (int bb = 0; bb < 4; ++bb) {
                float accx = wx[0] * fetch0(src, p, iz - 1 + c, iy - 1 + bb, ix - 1);
                accx = fmaf(wx[1], fetch0(src, p, iz - 1 + c, iy - 1 + bb, ix), accx);
                accx = fmaf(wx[2], fetch0(src, p, iz - 1 + c, iy - 1 + bb, ix + 1), accx);
                accx = fmaf(wx[3], fetch0(src, p, iz - 1 + c, iy - 1 + bb, ix + 2), accx);
                accy = fmaf(wy[bb], accx, accy);
            }
            val = fmaf(wz[c], accy, val);
        }
        return val;
    }
}

}  // namespace vt
