// vt_kernels_affine.hip -- the transform kernels, hand-written for gfx950 (MI355X, CDNA4).
//
// What the reference does per output voxel (transforms.py:253-281 + helper_interpolation.h): decode the
// linear index by div/mod, three float32 dot products with a matrix read from global memory, a skirt
// test, and one (linear), eight (cubicTex3D) or sixty-four (cubicTex3DSimple) fetches through the CUDA
// texture unit, which supplies the zero border, the trilinear blend and a 2-D/3-D-local cache.
//
// There is no texture unit on CDNA4, so the design is different:
//   * The output is cut into TD x TH x TW tiles, one 256-thread workgroup each.  The tile's source
//     footprint is a parallelepiped; its axis-aligned bounding box (plus the interpolation halo: 1 voxel
//     for linear, 2 for cubic) is staged once into LDS with coalesced 16-byte loads.  Source positions
//     outside the volume are staged as 0, which *is* the texture's border mode, so the gather needs no
//     per-tap bounds tests.
//   * Every lane then gathers its taps from LDS (8 for trilinear, 64 for the cubic B-spline) and
//     writes its voxel; lanes run along the fastest output axis so stores are coalesced.
//   * The 3x4 matrix, tile geometry and valid interval live in the kernarg segment -> SGPRs.
//   * Coordinates are evaluated in float64 from a per-tile base (exact for ordinary matrices), then
//     split into an integer LDS index and a float32 fraction; weights and sums are float32 as in
//     the reference's device functions.
//   * blockIdx -> tile mapping is XCD-aware: each of the 8 XCDs (private 4 MiB L2) walks a contiguous
//     range of tiles so neighbouring tiles' overlapping source boxes hit the same L2.
//
// A second, untiled kernel (`affine_direct`) gathers straight from global memory.  It is used for tiny
// volumes (launch-latency regime) and for matrices whose footprint does not fit LDS (large minification).
#include "vt_internal.h"
#include "vt_device.h"

namespace vt {

// Stage the source box [o, o+L) into LDS with direct-to-LDS loads (no VGPR round trip, every load of the
// workgroup in flight at once).  The LDS image is lane-linear: 16-byte vector v of the box lands at
// lds + 16*v.  Vectors outside the volume are fetched from a 16-byte block of zeros instead, which
// implements the texture unit's border mode without a second code path.
__device__ __forceinline__ void stage_box(float* lds, const float* __restrict__ src, const float* __restrict__ zeros16,
                                          const AffineParams& p, const int (&o)[3], int Lz, int Ly, int Lx, int tid)
{
    const int nvx = Lx >> 2;
    const int total = Lz * Ly * nvx;
    const int step_rows = 256 / nvx, step_cx = 256 - step_rows * nvx;
    const int step_z = step_rows / Ly, step_y = step_rows - step_z * Ly;
    int v = tid;
    int row = v / nvx;
    int cx = v - row * nvx;
    int z = row / Ly;
    int y = row - z * Ly;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    for (int vb = wave_first; vb < total; vb += 256, v += 256) {
        const int gz = o[0] + z, gy = o[1] + y, gx = o[2] + 4 * cx;
        const bool inb = (unsigned)gz < (unsigned)p.sD && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sP;
        const float* g = inb ? src + (((int64_t)gz * p.sH + gy) * p.sP + gx) : zeros16;
        if (v < total)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds + 4 * vb), 16, 0, 0);
        cx += step_cx; y += step_y; z += step_z;
        if (cx >= nvx) { cx -= nvx; y += 1; }
        if (y >= Ly) { y -= Ly; z += 1; }
    }
}

// One output voxel from the staged box.  (iz,iy,ix) = integer tap origin in box coordinates, f* = fractions.
template <int KIND>
__device__ __forceinline__ float sample_box(const float* __restrict__ lds, int Lx, int LyLx,
                                            int iz, int iy, int ix, float fz, float fy, float fx)
{
    if constexpr (KIND == 0) {
        const float* q = lds + (__mul24(iz, LyLx) + __mul24(iy, Lx) + ix);
        const float a000 = q[0], a001 = q[1];
        const float a010 = q[Lx], a011 = q[Lx + 1];
        const float a100 = q[LyLx], a101 = q[LyLx + 1];
        const float a110 = q[LyLx + Lx], a111 = q[LyLx + Lx + 1];
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
#ifndef VT_CUBIC_B32
        const int x1 = ix - 1, par = x1 & 1, e = x1 - par;
        const unsigned qa = lds_byte_address(lds + (__mul24(iz - 1, LyLx) + __mul24(iy - 1, Lx) + e));
        const unsigned row_b = 4u * (unsigned)Lx, plane_b = 4u * (unsigned)LyLx;
        return cubic_gather_b64([&](int c, int bb) { return qa + (unsigned)c * plane_b + (unsigned)bb * row_b; }, par, wx, wy, wz);
#else
        const float* q = lds + (__mul24(iz - 1, LyLx) + __mul24(iy - 1, Lx) + (ix - 1));
        float val = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float accy = 0.f;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float* rowp = q + c * LyLx + bb * Lx;
                float accx = wx[0] * rowp[0];
                accx = fmaf(wx[1], rowp[1], accx);
                accx = fmaf(wx[2], rowp[2], accx);
                accx = fmaf(wx[3], rowp[3], accx);
                accy = fmaf(wy[bb], accx, accy);
            }
            val = fmaf(wz[c], accy, val);
        }
        return val;
#endif
    }
}

template <int KIND /*0 linear, 1 cubic (bspline_weights), 2 cubic (bspline fn)*/, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void affine_tiled(const float* __restrict__ src, float* __restrict__ out,
                                                     const float* __restrict__ zeros16, const AffineParams p)
{
    static_assert(256 % TW == 0 && TH % (256 / TW) == 0, "tile/thread mapping");
    constexpr bool CUBIC = KIND != 0;
    constexpr int HALO = CUBIC ? 1 : 0;      // taps start at floor(s)-HALO
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tw_i = t % p.nTw;
    const int t2 = t / p.nTw;
    const int th_i = t2 % p.nTh;
    const int td_i = t2 / p.nTh;
    const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;

    // ---- tile geometry (wave-uniform, float64) ----
    double base[3], lo[3], hi[3];
    bool any_valid = true, all_valid = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        base[r] = fma(p.m[4 * r], (double)d0, fma(p.m[4 * r + 1], (double)h0, fma(p.m[4 * r + 2], (double)w0, p.m[4 * r + 3])));
        lo[r] = base[r] + p.neg[r];
        hi[r] = base[r] + p.pos[r];
        any_valid = any_valid && (hi[r] >= p.vlo[r] - kTileMargin) && (lo[r] < p.vhi[r] + kTileMargin);
        all_valid = all_valid && (lo[r] >= p.vlo[r] + kTileMargin) && (hi[r] < p.vhi[r] - kTileMargin);
    }

    const int kw = tid % TW;
    const int jh0 = tid / TW;
    constexpr int RP = 256 / TW;          // tile rows covered per pass
    constexpr int NJ = TH / RP;
    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;
    const int64_t ostride = (int64_t)p.oH * p.oW;
    const int nd = min(TD, p.oD - d0);

    if (!any_valid) {
        // the whole tile maps outside the valid interval: zero-fill (or leave untouched) and leave
        if (!keep) {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int h = h0 + jh0 + jj * RP, w = w0 + kw;
                if (h < p.oH && w < p.oW) {
                    float* optr = out + ((int64_t)d0 * p.oH + h) * p.oW + w;
                    for (int i = 0; i < nd; ++i, optr += ostride) *optr = 0.0f;
                }
            }
        }
        return;
    }

    // integer origin of the staged box (finite and small here: the tile intersects the valid interval and
    // its extent was bounded on the host).  x origin aligned down to 16 bytes: rows of the resident source
    // are 16-byte aligned (pitch % 4 == 0).
    int o[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = (int)floor(lo[r]) - HALO;
    o[2] &= ~3;

    const int Lx = p.Lx, Ly = p.Ly, Lz = p.Lz;
    stage_box(lds, src, zeros16, p, o, Lz, Ly, Lx, tid);
    __syncthreads();     // hipcc drains the direct-to-LDS loads (vmcnt(0)) ahead of the barrier

    // ---- gather ----
    // box-relative coordinate of output voxel (i,j,k): rel_r = (base_r - o_r) + m[r][0]*i + m[r][1]*j + m[r][2]*k
    double b[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) b[r] = base[r] - (double)o[r];
    const int LyLx = Ly * Lx;

#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = jh0 + jj * RP;
        const int h = h0 + j, w = w0 + kw;
        if (h >= p.oH || w >= p.oW) continue;
        const double s0 = fma(p.m[1], (double)j, fma(p.m[2], (double)kw, b[0]));
        const double s1 = fma(p.m[5], (double)j, fma(p.m[6], (double)kw, b[1]));
        const double s2 = fma(p.m[9], (double)j, fma(p.m[10], (double)kw, b[2]));
        Fx c0 = to_fx(s0), c1 = to_fx(s1), c2 = to_fx(s2);
        float* optr = out + ((int64_t)d0 * p.oH + h) * p.oW + w;
        if (all_valid && nd == TD) {
            // interior tile: no per-voxel tests, unrolled so several voxels' LDS reads are in flight
#pragma unroll 4
            for (int i = 0; i < TD; ++i) {
                optr[i * ostride] = sample_box<KIND>(lds, Lx, LyLx, c0.hi, c1.hi, c2.hi, fx_frac(c0), fx_frac(c1), fx_frac(c2));
                fx_step(c0, p.inc_hi[0], p.inc_lo[0]);
                fx_step(c1, p.inc_hi[1], p.inc_lo[1]);
                fx_step(c2, p.inc_hi[2], p.inc_lo[2]);
            }
        } else {
            // tiles cut by the skirt or by the end of the output: the inside test is done on float64
            // coordinates (identical to the direct kernel and the oracle), the taps still come from the
            // fixed-point split
            for (int i = 0; i < nd; ++i) {
                const bool inside = canonical_inside(p, d0 + i, h, w);
                const float val = sample_box<KIND>(lds, Lx, LyLx, c0.hi, c1.hi, c2.hi, fx_frac(c0), fx_frac(c1), fx_frac(c2));
                if (inside) optr[i * ostride] = val;
                else if (!keep) optr[i * ostride] = 0.0f;
                fx_step(c0, p.inc_hi[0], p.inc_lo[0]);
                fx_step(c1, p.inc_hi[1], p.inc_lo[1]);
                fx_step(c2, p.inc_hi[2], p.inc_lo[2]);
            }
        }
    }
}

#ifdef VT_LEGACY      // kernel 3 (box kernel with in-plane partial reuse for axis-0-separable matrices): test build only
// ---------------------------------------------------------------------------------------------------
// axis-0-separable tiled kernel
// ---------------------------------------------------------------------------------------------------
// When the matrix has the block form  [1 0 0 tz; 0 a b ty; 0 c d tx]  -- rotations about axis 0 (the
// README sweep `rotate((0, i, 0))`, every BASELINE configuration), in-plane scale/shear, any translation --
// the source plane of an output voxel depends only on d and its in-plane position only on (h, w):
//   * (iy, ix, fy, fx) and the in-plane weights are computed once per thread and reused for all TD planes;
//   * the z fraction is the same for every voxel of the launch (wave-uniform: it lives in an SGPR);
//   * consecutive output planes read consecutive source planes, so the in-plane partial result of a source
//     plane (a bilinear blend, or the 16-tap in-plane B-spline sum) is computed once and reused by the 2
//     (linear) or 4 (cubic) output planes that need it -- 4 (16) LDS reads per voxel instead of 8 (64).
// The arithmetic and its association (x, then y, then z) are exactly those of the general kernel.
template <int KIND>
__device__ __forceinline__ float plane_partial(const float* __restrict__ q, int Lx, float fy, float fx,
                                               const float (&wy)[4], const float (&wx)[4])
{
    if constexpr (KIND == 0) {
        const float a00 = q[0], a01 = q[1], a10 = q[Lx], a11 = q[Lx + 1];
        const float x0 = fmaf(fx, a01 - a00, a00);
        const float x1 = fmaf(fx, a11 - a10, a10);
        return fmaf(fy, x1 - x0, x0);
    } else {
        float accy = 0.f;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float* rowp = q + bb * Lx;
            float accx = wx[0] * rowp[0];
            accx = fmaf(wx[1], rowp[1], accx);
            accx = fmaf(wx[2], rowp[2], accx);
            accx = fmaf(wx[3], rowp[3], accx);
            accy = fmaf(wy[bb], accx, accy);
        }
        return accy;
    }
}

template <int KIND, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void affine_tiled_zsep(const float* __restrict__ src, float* __restrict__ out,
                                                          const float* __restrict__ zeros16, const AffineParams p)
{
    static_assert(256 % TW == 0 && TH % (256 / TW) == 0, "tile/thread mapping");
    constexpr bool CUBIC = KIND != 0;
    constexpr int HALO = CUBIC ? 1 : 0;
    constexpr int NPL = TD + 1 + 2 * HALO;        // source planes staged per tile
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tw_i = t % p.nTw;
    const int t2 = t / p.nTw;
    const int th_i = t2 % p.nTh;
    const int td_i = t2 / p.nTh;
    const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
    const int nd = min(TD, p.oD - d0);

    // in-plane footprint of the tile (rows 1, 2 of the matrix; column 0 is zero)
    double base[3], lo[3], hi[3];
    bool any_valid = true, all_valid = true;
#pragma unroll
    for (int r = 1; r < 3; ++r) {
        base[r] = fma(p.m[4 * r + 1], (double)h0, fma(p.m[4 * r + 2], (double)w0, p.m[4 * r + 3]));
        lo[r] = base[r] + p.neg[r];
        hi[r] = base[r] + p.pos[r];
        any_valid = any_valid && (hi[r] >= p.vlo[r] - kTileMargin) && (lo[r] < p.vhi[r] + kTileMargin);
        all_valid = all_valid && (lo[r] >= p.vlo[r] + kTileMargin) && (hi[r] < p.vhi[r] - kTileMargin);
    }
    // axis 0: src_z = d + tz exactly; zoff = floor(tz), fz = tz - zoff (host-computed)
    const double z_lo = (double)d0 + p.m[3], z_hi = (double)(d0 + nd - 1) + p.m[3];
    any_valid = any_valid && (z_hi >= p.vlo[0] - kTileMargin) && (z_lo < p.vhi[0] + kTileMargin);
    all_valid = all_valid && (z_lo >= p.vlo[0] + kTileMargin) && (z_hi < p.vhi[0] - kTileMargin) && (nd == TD);

    const int kw = tid % TW;
    const int jh0 = tid / TW;
    constexpr int RP = 256 / TW;
    constexpr int NJ = TH / RP;
    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;
    const int64_t ostride = (int64_t)p.oH * p.oW;

    if (!any_valid) {
        if (!keep) {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int h = h0 + jh0 + jj * RP, w = w0 + kw;
                if (h < p.oH && w < p.oW) {
                    float* optr = out + ((int64_t)d0 * p.oH + h) * p.oW + w;
                    for (int i = 0; i < nd; ++i, optr += ostride) *optr = 0.0f;
                }
            }
        }
        return;
    }

    int o[3];
    o[0] = d0 + p.zoff - HALO;
    o[1] = (int)floor(lo[1]) - HALO;
    o[2] = ((int)floor(lo[2]) - HALO) & ~3;
    const int Lx = p.Lx, Ly = p.Ly;
    stage_box(lds, src, zeros16, p, o, NPL, Ly, Lx, tid);
    __syncthreads();

    const int LyLx = Ly * Lx;
    const float fz = p.fz;
    float wz[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CUBIC) cubic_weights<KIND == 2>(fz, wz);
    const double by = base[1] - (double)o[1], bx = base[2] - (double)o[2];

#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = jh0 + jj * RP;
        const int h = h0 + j, w = w0 + kw;
        if (h >= p.oH || w >= p.oW) continue;
        const double sy = fma(p.m[5], (double)j, fma(p.m[6], (double)kw, by));
        const double sx = fma(p.m[9], (double)j, fma(p.m[10], (double)kw, bx));
        const double fyd = floor(sy), fxd = floor(sx);
        const int iy = (int)fyd, ix = (int)fxd;
        const float fy = (float)(sy - fyd), fx = (float)(sx - fxd);
        float wy[4] = {0.f, 0.f, 0.f, 0.f}, wx[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (CUBIC) { cubic_weights<KIND == 2>(fy, wy); cubic_weights<KIND == 2>(fx, wx); }
        bool in_yx = true;
        if (!all_valid) in_yx = canonical_inside_axis(p, 1, d0, h, w) && canonical_inside_axis(p, 2, d0, h, w);   // rows 1, 2 ignore d
        const float* q = lds + (__mul24(iy - HALO, Lx) + (ix - HALO));
        float* optr = out + ((int64_t)d0 * p.oH + h) * p.oW + w;

        if constexpr (!CUBIC) {
            float b0 = plane_partial<KIND>(q, Lx, fy, fx, wy, wx);
#pragma unroll 4
            for (int i = 0; i < TD; ++i) {
                q += LyLx;
                const float b1 = plane_partial<KIND>(q, Lx, fy, fx, wy, wx);
                const float val = fmaf(fz, b1 - b0, b0);
                b0 = b1;
                if (all_valid) optr[i * ostride] = val;
                else if (i < nd) {
                    const double ez = (double)(d0 + i) + p.m[3];
                    const bool inside = in_yx && (ez >= p.vlo[0]) && (ez < p.vhi[0]);
                    if (inside) optr[i * ostride] = val;
                    else if (!keep) optr[i * ostride] = 0.0f;
                }
            }
        } else {
            float p0 = plane_partial<KIND>(q, Lx, fy, fx, wy, wx);
            float p1 = plane_partial<KIND>(q + LyLx, Lx, fy, fx, wy, wx);
            float p2 = plane_partial<KIND>(q + 2 * LyLx, Lx, fy, fx, wy, wx);
            q += 3 * LyLx;
#pragma unroll 4
            for (int i = 0; i < TD; ++i, q += LyLx) {
                const float p3 = plane_partial<KIND>(q, Lx, fy, fx, wy, wx);
                float val = wz[0] * p0;          // = fma(wz[0], p0, 0): same association as the general kernel
                val = fmaf(wz[1], p1, val);
                val = fmaf(wz[2], p2, val);
                val = fmaf(wz[3], p3, val);
                p0 = p1; p1 = p2; p2 = p3;
                if (all_valid) optr[i * ostride] = val;
                else if (i < nd) {
                    const double ez = (double)(d0 + i) + p.m[3];
                    const bool inside = in_yx && (ez >= p.vlo[0]) && (ez < p.vhi[0]);
                    if (inside) optr[i * ostride] = val;
                    else if (!keep) optr[i * ostride] = 0.0f;
                }
            }
        }
    }
}

#endif  // VT_LEGACY

// ---------------------------------------------------------------------------------------------------
// direct kernel: one thread per output voxel, taps from global memory with explicit border tests
// ---------------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void affine_direct(const float* __restrict__ src, float* __restrict__ out,
                                                      const AffineParams p)
{
    const int64_t n = (int64_t)p.oD * p.oH * p.oW;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int w = (int)(idx % p.oW);
    const int64_t r2 = idx / p.oW;
    const int h = (int)(r2 % p.oH);
    const int d = (int)(r2 / p.oH);

    double s[3];
    bool inside = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        s[r] = fma(p.m[4 * r], (double)d, fma(p.m[4 * r + 1], (double)h, fma(p.m[4 * r + 2], (double)w, p.m[4 * r + 3])));
        inside = inside && (s[r] >= p.vlo[r]) && (s[r] < p.vhi[r]);
    }
    if (!inside) {
        if (!(p.flags & VT_KEEP_OUTSIDE)) out[idx] = 0.0f;
        return;
    }
    const double fzd = floor(s[0]), fyd = floor(s[1]), fxd = floor(s[2]);
    const int iz = (int)fzd, iy = (int)fyd, ix = (int)fxd;
    const float fz = (float)(s[0] - fzd), fy = (float)(s[1] - fyd), fx = (float)(s[2] - fxd);
    const float val = direct_sample<KIND>(src, p, iz, iy, ix, fz, fy, fx);
    out[idx] = val;
}

// Batch form for small volumes (launch-latency regime: one launch serves `n` matrices; blockIdx.y = matrix).
// ms: n x 12 doubles (3x4 pull matrices with the plane offsets folded in); out: n consecutive volumes.
template <int KIND>
__global__ __launch_bounds__(256) void affine_direct_batch(const float* __restrict__ src, float* __restrict__ out,
                                                            const double* __restrict__ ms, const AffineParams p)
{
    const int64_t n = (int64_t)p.oD * p.oH * p.oW;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const double* m = ms + 12 * (int64_t)blockIdx.y;              // wave-uniform: scalar loads
    float* o = out + (int64_t)blockIdx.y * n;
    const int w = (int)(idx % p.oW);
    const int64_t r2 = idx / p.oW;
    const int h = (int)(r2 % p.oH);
    const int d = (int)(r2 / p.oH);
    double s[3];
    bool inside = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        s[r] = fma(m[4 * r], (double)d, fma(m[4 * r + 1], (double)h, fma(m[4 * r + 2], (double)w, m[4 * r + 3])));
        inside = inside && (s[r] >= p.vlo[r]) && (s[r] < p.vhi[r]);
    }
    if (!inside) {
        if (!(p.flags & VT_KEEP_OUTSIDE)) o[idx] = 0.0f;
        return;
    }
    const double fzd = floor(s[0]), fyd = floor(s[1]), fxd = floor(s[2]);
    o[idx] = direct_sample<KIND>(src, p, (int)fzd, (int)fyd, (int)fxd, (float)(s[0] - fzd), (float)(s[1] - fyd), (float)(s[2] - fxd));
}

// ---------------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------------
struct TileCfg { int td, th, tw; };
static const TileCfg kTiles[] = {
    {16, 16, 16},   // 0: cube -- smallest box for general 3-D rotations
    {8, 16, 32},    // 1: 128-byte store segments
    {8, 16, 16},    // 2: half cube -- twice the workgroups per CU
    {8, 8, 32},     // 3: half-size for fat footprints
    {4, 8, 32},     // 4: minification up to ~3x
};

int tile_config_count() { return (int)(sizeof(kTiles) / sizeof(kTiles[0])); }
void tile_config(int idx, int* td, int* th, int* tw) { *td = kTiles[idx].td; *th = kTiles[idx].th; *tw = kTiles[idx].tw; }

typedef void (*tiled_fn)(const float*, float*, const float*, const AffineParams);

template <int TD, int TH, int TW>
static tiled_fn pick_tiled(int kind, bool zsep)
{
#ifdef VT_LEGACY
    if (zsep) {
        switch (kind) {
            case 0: return affine_tiled_zsep<0, TD, TH, TW>;
            case 1: return affine_tiled_zsep<1, TD, TH, TW>;
            default: return affine_tiled_zsep<2, TD, TH, TW>;
        }
    }
#endif
    switch (kind) {
        case 0: return affine_tiled<0, TD, TH, TW>;
        case 1: return affine_tiled<1, TD, TH, TW>;
        default: return affine_tiled<2, TD, TH, TW>;
    }
}

static tiled_fn tiled_entry(int cfg, int kind, bool zsep)
{
    switch (cfg) {
        case 0: return pick_tiled<16, 16, 16>(kind, zsep);
        case 1: return pick_tiled<8, 16, 32>(kind, zsep);
        case 2: return pick_tiled<8, 16, 16>(kind, zsep);
        case 3: return pick_tiled<8, 8, 32>(kind, zsep);
        default: return pick_tiled<4, 8, 32>(kind, zsep);
    }
}

int interp_kind(int interp)
{
    switch (interp) {
        case VT_LINEAR: return 0;
        case VT_BSPLINE: case VT_FILT_BSPLINE: return 1;
        default: return 2;
    }
}

hipError_t init_affine_kernels()
{
    // dynamic LDS above 64 KiB needs an explicit opt-in per kernel; gfx950 has 160 KiB per workgroup
    for (int cfg = 0; cfg < tile_config_count(); ++cfg)
        for (int kind = 0; kind < 3; ++kind)
#ifdef VT_LEGACY
            for (int z = 0; z < 2; ++z) {
#else
            for (int z = 0; z < 1; ++z) {
#endif
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tiled_entry(cfg, kind, z != 0)),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
#ifdef VT_LEGACY
    return init_march_kernels();
#else
    return hipSuccess;
#endif
}

hipError_t launch_affine_tiled(int cfg, int interp, bool zsep, const float* src, float* out, const float* zeros16,
                               const AffineParams& p, int grid, int lds_bytes, hipStream_t stream)
{
    tiled_fn fn = tiled_entry(cfg, interp_kind(interp), zsep);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds_bytes, stream, src, out, zeros16, p);
    return hipGetLastError();
}

hipError_t launch_affine_direct(int interp, const float* src, float* out, const AffineParams& p,
                                hipStream_t stream)
{
    const int64_t n = (int64_t)p.oD * p.oH * p.oW;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    switch (interp_kind(interp)) {
        case 0: hipLaunchKernelGGL(affine_direct<0>, dim3((unsigned)blocks), dim3(256), 0, stream, src, out, p); break;
        case 1: hipLaunchKernelGGL(affine_direct<1>, dim3((unsigned)blocks), dim3(256), 0, stream, src, out, p); break;
        default: hipLaunchKernelGGL(affine_direct<2>, dim3((unsigned)blocks), dim3(256), 0, stream, src, out, p); break;
    }
    return hipGetLastError();
}

hipError_t launch_affine_direct_batch(int interp, const float* src, float* out, const double* d_ms, int n,
                                      const AffineParams& p, hipStream_t stream)
{
    const int64_t nvox = (int64_t)p.oD * p.oH * p.oW;
    const int64_t blocks = (nvox + 255) / 256;
    if (blocks > 0x7fffffffLL || n <= 0 || n > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks, (unsigned)n);
    switch (interp_kind(interp)) {
        case 0: hipLaunchKernelGGL(affine_direct_batch<0>, grid, dim3(256), 0, stream, src, out, d_ms, p); break;
        case 1: hipLaunchKernelGGL(affine_direct_batch<1>, grid, dim3(256), 0, stream, src, out, d_ms, p); break;
        default: hipLaunchKernelGGL(affine_direct_batch<2>, grid, dim3(256), 0, stream, src, out, d_ms, p); break;
    }
    return hipGetLastError();
}

}  // namespace vt
