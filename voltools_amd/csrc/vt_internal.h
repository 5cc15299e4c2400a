// vt_internal.h -- shared declarations of the HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/voltools_hip.h"

namespace vt {

// Prefilter constants in float32, evaluated the way the reference's device code does
// (helper_math.h:1468 `Pole = sqrt(3.0f)-2.0f`; bspline.h:36 Lambda; bspline.h:27 anticausal gain).
constexpr float kPole   = -0x1.126148p-2f;   // -0.26794922
constexpr float kLambda =  0x1.7ffffep+2f;   //  5.9999995 = (1-Pole)*(1-1/Pole)
constexpr float kAntiInit = 0x1.b0cb1ap-3f;  //  0.21132489 = Pole/(Pole-1)

// Kernel argument block of the transform kernels.  Lives in the kernarg segment, i.e. it is read with
// scalar loads into SGPRs once per wave ("matrix broadcast from constant memory").
struct AffineParams {
    double m[12];      // 3x4 pull matrix in array-axis order; output-plane and slab offsets folded into m[r][3]
    double neg[3];     // sum_c min(0, m[r][c]*(T_c-1)): lowest source coordinate of a tile relative to its base
    double pos[3];     // sum_c max(0, m[r][c]*(T_c-1))
    double vlo[3];     // valid source interval [vlo, vhi) per axis: the skirt rule src+0.5 in [0, dim)
    double vhi[3];
    int32_t inc_hi[3];         // Q32.32 split of m[r][0]: per-step increment along the tile's depth axis
    uint32_t inc_lo[3];
    int32_t sD, sH, sW;        // resident source dims
    int32_t sP;                // row pitch of the resident source in floats (multiple of 4, pad columns are 0)
    int32_t oD, oH, oW;        // output dims
    int32_t nTd, nTh, nTw;     // output tile counts
    int32_t Lz, Ly, Lx;        // staged source box (floats); Lx is the LDS row stride
    int32_t flags;             // VT_KEEP_OUTSIDE
    int32_t zoff;              // axis-0-separable launches: src_z = d + zoff + fz
    float fz;
    int32_t dch;               // marching kernel: output planes per workgroup
    int32_t zero_off;          // byte offset, inside any source plane, of a 16-byte vector of zeros (row pad)
    int32_t slot_floats;       // marching kernel: floats per LDS plane slot (packed footprint, multiple of 4)
    int32_t sP2;               // plane-pair layout: floats per pair-row (2 * (roundup4(W) + 4))
    int32_t zero_off2;         // plane-pair layout: byte offset of a zero vector inside any pair-plane
    double ia1, ib1;           // marching kernels: march_recip(m[1][1]), march_recip(m[1][2])
    int32_t Lx_used;           // marching kernels, box mode: columns of a staged row that hold data (the rest of Lx is padding)
    int32_t ord[3];            // column order of the skirt test's fma chain (0,1,2; permuted when the launch runs on an axis-exchanged copy,
                               // so that the chain is the original problem's and boundary voxels classify as in affine_direct)
    int64_t ostride, orow;     // marching kernels: element stride between output planes / rows (oH*oW, oW unless axes are swapped)
    int blk_h, blk_w;          // marching kernels: blocked tile order inside a chunk layer (tiles per block; 0 = plain order)
    int32_t sPq;               // plane-quad layout: floats per quad-row (4 * positions per row)
    int32_t zero_off_q;        // plane-quad layout: byte offset of a zero vector inside any quad-plane
    uint32_t nTw_magic, nTh_magic;   // plane-quad kernel, 2-D grid: floor(2^32 / n) + 1 -- (u * magic) >> 32 == u / n for u * n < 2^32
    int32_t row_s;             // plane-quad kernel: bank-aware row starts, slot = column + row * row_s (mod 16); -1 = rows packed back to back
    int32_t dshift;            // plane-quad kernel: chunk c > 0 starts at output plane c*dch + dshift (0..3), chosen so that a chunk's first
                               // tap plane is the first plane of a quad (one quad step per chunk beyond its own planes instead of two)
    int32_t Lps;               // block kernel: LDS plane stride in floats (Ly * row stride + bank padding, multiple of 4)
    uint32_t psv_magic;        // block kernel: floor(2^32 / (Lps / 4)) + 1
    int32_t lds_cap;           // block kernel: bytes of LDS the box may take (the tile-queue word follows)
    int32_t binc_hi[5][3];     // block kernel: Q32.32 increments of the steps between a thread's eight voxels (+8 w, +8 h, -8 w, +4 d, -8 h),
    uint32_t binc_lo[5][3];    // [step kind][source axis]
};


// Columns [*mn, *mx] (box-relative) that the pixels of a TH x TW in-plane tile tap in box row Y, for the in-plane map
// sy = by + a1*j + b1*k, sx = bx + a2*j + b2*k.  A pixel taps the row iff sy lies in [Y-1-halo, Y+halo+1); over the
// continuous pixel rectangle that is a convex polygon and the extreme sx over it is attained at a vertex: a rectangle
// corner inside the strip, or a point where a strip line crosses a rectangle edge.  The result is a superset of the
// taps of the discrete pixels (widened by 1e-6).  Shared by the marching kernel (per workgroup) and the host planner
// (slot sizing), so both see the same spans.  Returns false when no pixel taps the row.
// ia1 / ib1: reciprocals of a1 / b1, 0 where the coefficient vanishes (march_recip; wave-uniform, computed on the host).
__host__ __device__ inline double march_recip(double a) { return (a > 1e-12 || a < -1e-12) ? 1.0 / a : 0.0; }

__host__ __device__ inline bool march_row_span(double a1, double b1, double a2, double b2, double ia1, double ib1,
                                               double by, double bx, int Y, int TH, int TW, int halo, int* mn, int* mx)
{
    const double ylo = (double)(Y - 1 - halo), yhi = (double)(Y + halo + 1);
    const double jm = (double)(TH - 1), km = (double)(TW - 1);
    double smin = 1e30, smax = -1e30;
    for (int cj = 0; cj < 2; ++cj)
        for (int ck = 0; ck < 2; ++ck) {
            const double j = cj ? jm : 0.0, k = ck ? km : 0.0;
            const double sy = by + a1 * j + b1 * k;
            if (sy >= ylo - 1e-6 && sy <= yhi + 1e-6) {
                const double sx = bx + a2 * j + b2 * k;
                smin = sx < smin ? sx : smin;
                smax = sx > smax ? sx : smax;
            }
        }
    for (int e = 0; e < 2; ++e) {
        const double L = e ? yhi : ylo;
        for (int c = 0; c < 2; ++c) {
            if (ib1 != 0.0) {                 // edges j = 0 and j = TH-1: solve for k
                const double j = c ? jm : 0.0;
                const double k = (L - by - a1 * j) * ib1;
                if (k >= 0.0 && k <= km) {
                    const double sx = bx + a2 * j + b2 * k;
                    smin = sx < smin ? sx : smin;
                    smax = sx > smax ? sx : smax;
                }
            }
            if (ia1 != 0.0) {                 // edges k = 0 and k = TW-1: solve for j
                const double k = c ? km : 0.0;
                const double j = (L - by - b1 * k) * ia1;
                if (j >= 0.0 && j <= jm) {
                    const double sx = bx + a2 * j + b2 * k;
                    smin = sx < smin ? sx : smin;
                    smax = sx > smax ? sx : smax;
                }
            }
        }
    }
    if (!(smin <= smax)) { *mn = 0; *mx = -1; return false; }
    *mn = (int)floor(smin - 1e-6) - halo;
    *mx = (int)floor(smax + 1e-6) + 1 + halo;
    return true;
}

// Geometry of a tile's source footprint for the packed general-matrix kernel: u' = A.p - neg is the position of tile
// voxel p inside the footprint's bounding box (before the sub-voxel offset of the tile is added).
struct PackGeom {
    double inv[9];     // A^-1 (rows: tile axes d,h,w; columns: source axes z,y,x)
    double cst[3];     // A^-1 . neg
    double ext[3];     // bounding-box extent of A.[0,T-1]^3 per source axis
    int32_t T[3];      // tile dims
    int32_t halo;      // 0 linear, 1 cubic
    int32_t Lxbox;     // bounding-box row length (floats) incl. alignment slack
    int32_t Lybox;
};

struct TilePlan {
    int kind;            // 1 direct, 2 tiled, 3 tiled axis-0-separable, 4 marching, 5 marching on plane pairs, 6 tiled with packed footprints,
                         // 7 reserved (vt_volume_info: fused projection), 8 marching on plane quads, 9 lane-block tiles (cfg = row-stride index),
                         // 10 source rows along w (maps that leave axis 2 alone)
    int cfg;             // index into the tile table
    int td, th, tw;
    int lds_bytes;
    int grid;
    int blocks_per_cu;
    PackGeom geo;       // kind 6 only
};

// Columns [*mn, *mx] (box-relative) that tile voxels can tap in box row (Z, Y), for EVERY sub-voxel position of a tile:
// box coordinate = b + u' with b_z, b_y in [halo, halo+1), b_x in [halo, halo+4) (16-byte alignment of the box origin).
// Row Z is tapped iff u'_z in (Z-2-2*halo, Z+1), same for Y.  Each pair of opposite faces of the tile bounds u'_x by an
// interval that is linear in (u'_z, u'_y); over the rectangle of admissible (u'_z, u'_y) the lower end of the feasible u'_x
// is >= max over faces of (min over the 4 corners), the upper end <= min over faces of (max over corners).
// Conservative (a superset of the taps), cheap (3 x 4 corner evaluations), identical on host and device.
__host__ __device__ inline bool packed_row_span(const PackGeom& g, int Z, int Y, int* mn, int* mx)
{
    const double zlo = (double)(Z - 2 - 2 * g.halo) - 1e-6, zhi = (double)(Z + 1) + 1e-6;
    const double ylo = (double)(Y - 2 - 2 * g.halo) - 1e-6, yhi = (double)(Y + 1) + 1e-6;
    // rows the footprint cannot reach at all
    if (zhi < 0.0 || zlo > g.ext[0] || yhi < 0.0 || ylo > g.ext[1]) { *mn = 0; *mx = -1; return false; }
    double LB = 0.0, UB = g.ext[2];
    for (int c = 0; c < 3; ++c) {
        const double gz = g.inv[3 * c], gy = g.inv[3 * c + 1], gx = g.inv[3 * c + 2];
        if (gx > -1e-9 && gx < 1e-9) continue;                  // this pair of faces does not bound x
        const double inv_gx = 1.0 / gx;
        const double top = (double)(g.T[c] - 1);
        double lo_min = 1e30, hi_max = -1e30;
        for (int k = 0; k < 4; ++k) {
            const double uz = (k & 1) ? zhi : zlo, uy = (k & 2) ? yhi : ylo;
            const double s = gz * uz + gy * uy + g.cst[c];
            const double xa = (0.0 - s) * inv_gx, xb = (top - s) * inv_gx;
            const double lo = xa < xb ? xa : xb, hi = xa < xb ? xb : xa;
            lo_min = lo < lo_min ? lo : lo_min;
            hi_max = hi > hi_max ? hi : hi_max;
        }
        LB = lo_min > LB ? lo_min : LB;
        UB = hi_max < UB ? hi_max : UB;
    }
    if (LB > UB + 1e-6) { *mn = 0; *mx = -1; return false; }
    int a = (int)floor(LB - 1e-6);                               // + b_x (>= halo) - halo taps
    int b = (int)floor(UB + 1e-6 + (double)g.halo + 4.0) + 1 + g.halo;
    if (a < 0) a = 0;
    if (b > g.Lxbox - 1) b = g.Lxbox - 1;
    *mn = a;
    *mx = b;
    return a <= b;
}

// Axis-0 projection (vt_kernels_project.hip): dst[y, x] = sum_z c_z * src[z, y, x]
struct ProjectParams {
    int32_t D, H;              // planes to sum, rows per plane
    int32_t nxv;               // column groups per row (VEC floats each)
    int32_t vec;               // 4: float4 per lane (pitches multiples of 4), 1: scalar
    int32_t src_pitch;         // floats per source row
    int32_t dst_pitch;         // floats per destination row
    int64_t dst_plane;         // floats between the `copies` destination planes
    int32_t copies;            // the sum is written `copies` times (3 planes of the helper volume)
    int32_t uniform;           // 1: c_z = 1
    int32_t zoff, halo, ntap;  // output plane d taps source planes d + zoff - halo + k, k < ntap, with weight wz[k]
    int32_t dlo, dhi;          // valid output planes (inclusive)
    float wz[4];
};
hipError_t launch_plane_sum(const float* src, float* dst, const ProjectParams& q, hipStream_t stream);

// launchers (vt_kernels_affine.hip)
int tile_config_count();
void tile_config(int idx, int* td, int* th, int* tw);
hipError_t launch_affine_tiled(int cfg, int interp, bool zsep, const float* src, float* out, const float* zeros16,
                               const AffineParams& p, int grid, int lds_bytes, hipStream_t stream);
hipError_t launch_affine_direct_batch(int interp, const float* src, float* out, const double* d_ms, int n,
                                      const AffineParams& p, hipStream_t stream);
int march_table_bytes();   // LDS bytes the packed-span set-up table needs (overlays the ring)
int march_config_count();
void march_config(int idx, int* th, int* tw, int* g, int* la, int* nt);
int zpair_config_count();
void zpair_config(int idx, int* th, int* tw, int* la, int* nt);
int march_blocks_per_cu(bool pair, int cfg, int interp, int lds_bytes);   // resident workgroups per CU (occupancy calculator, cached)
hipError_t launch_affine_zpair(int cfg, int interp, const float* src2, float* out, const AffineParams& p,
                               int grid, int lds_bytes, hipStream_t stream);
hipError_t launch_relayout_zpair(const float* src, float* dst, int D, int H, int W, int P, int P2, hipStream_t stream);
// dst[k][j][i] = src[i][j][k]; element strides: src i*ss0 + j*ss1 + k, dst k*ds0 + j*ds1 + i (vt_kernels_layout.hip)
hipError_t launch_transpose02(const float* src, float* dst, int n0, int n1, int n2, int64_t ss0, int64_t ss1,
                              int64_t ds0, int64_t ds1, hipStream_t stream);
hipError_t launch_mirror_pad(const float* src, float* dst, int D, int H, int W, int R, int P, hipStream_t stream);
hipError_t launch_relayout_swap01(const float* src, float* dst, int D, int H, int P, hipStream_t stream);
// plane-quad marching kernel (vt_kernels_quad.hip)
int quad_max_it();
int quad_config_count();
void quad_config(int idx, int* th, int* tw, int* nt);
int quad_blocks_per_cu(int cfg, int interp, int lds_bytes, bool zid = false);
hipError_t init_quad_kernels();
hipError_t launch_relayout_zquad(const float* src, float* dst, int D, int H, int W, int P, int Pq, hipStream_t stream);
// vt_kernels_rows.hip (kind 10: maps that leave axis 2 alone)
hipError_t launch_relayout_xfir(const float* src, float* dst, int D, int H, int W, int P, bool simple, hipStream_t stream);
hipError_t init_rows_kernels();
void rows_tile(int* ph, int* run);
hipError_t launch_affine_rows(int interp, int pd, const float* src, float* out, const float* zeros16, const AffineParams& p, int lds_bytes, hipStream_t stream);
hipError_t launch_relayout_zquad_fir(const float* src, float* dst, int D, int H, int W, int P, int Pq, bool simple, hipStream_t stream);
// the same two forms of the in-plane transposed orientation, straight from the plain copy (no exchanged plain copy in between)
hipError_t launch_relayout_zquad_swap12(const float* src, float* dst, int D, int H, int W, int P, int Pq, bool fir, bool simple, hipStream_t stream);
hipError_t launch_affine_quad(int cfg, int interp, const float* srcq, float* out, const AffineParams& p,
                              int grid, int lds_bytes, hipStream_t stream);
// lane-block kernel for general matrices (vt_kernels_block.hip)
int block_rs_count();
int block_rs(int idx);
int block_rs_order(int th, const int** order);   // row-stride indices in the planner's order of preference for tile height th (16 or 8)
int block_max_vectors(int th);
void block_tile(int th, int* td, int* tw);
hipError_t init_block_kernels();
hipError_t launch_affine_block(int rs_idx, int th, int interp, const float* src, float* out, const float* zeros16, int* queue,
                               const AffineParams& p, const PackGeom& geo, int grid, int lds_bytes, hipStream_t stream);
int packed_config_count();
void packed_config(int idx, int* td, int* th, int* tw);
int packed_rows_max();
int packed_vectors_max();
hipError_t init_packed_kernels();
hipError_t launch_affine_packed(int cfg, int interp, const float* src, float* out, const float* zeros16, int* queue,
                                const AffineParams& p, const PackGeom& geo, int grid, int lds_bytes, hipStream_t stream);
// trilinear packed-span kernel of round 5 (vt_kernels_span.hip)
int span_config_count();
void span_config(int idx, int* td, int* th, int* tw);
int span_rows_max();
bool span_config_pipelined(int idx);      // two footprint buffers: LDS = table + 2 x capacity
int span_vectors_max();
hipError_t init_span_kernels();
hipError_t launch_affine_span(int cfg, const float* src, float* out, const float* zeros16, int* queue,
                              const AffineParams& p, const PackGeom& geo, int grid, int lds_bytes, hipStream_t stream);
int march_rows_max();
int march_max_it();
int interp_kind(int interp);
hipError_t init_march_kernels();
hipError_t launch_affine_march(int cfg, int interp, const float* src, float* out, const AffineParams& p,
                               int grid, int lds_bytes, hipStream_t stream);
hipError_t launch_affine_direct(int interp, const float* src, float* out, const AffineParams& p,
                                hipStream_t stream);
hipError_t init_affine_kernels();   // raises the dynamic-LDS limit of every tiled instantiation

// prefilter (vt_kernels_prefilter.hip).  src -> dst; `*in_place_ok` tells whether src == dst is legal.
// axis: 0 (Z, stride H*W), 1 (Y, stride W), 2 (X, contiguous).
hipError_t launch_prefilter_axis(int axis, const float* src, float* dst, int D, int H, int W, int pitch,
                                 bool lo_interior, hipStream_t stream);
int prefilter_chunk_size(int N);                  // samples per chunk of the strided passes on lines of N samples
int prefilter_warmup();                           // samples of warm-up on each side of a chunk
hipError_t launch_prefilter_axis0_chunks(const float* src, float* dst, int D, int H, int W, int pitch, int c0, int c1, hipStream_t stream);
bool prefilter_axis_in_place_ok(int axis, int D, int H, int W);
bool prefilter_xy_ok(int D, int H, int W, int pitch, const void* src, const void* dst);
hipError_t launch_prefilter_xy(const float* src, float* dst, int D, int H, int W, int pitch, hipStream_t stream);

}  // namespace vt
