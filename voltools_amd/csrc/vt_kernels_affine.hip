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

namespace vt {

// ---------------------------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------------------------

// bspline.h:102-112
__device__ __forceinline__ void bspline_weights(float f, float& w0, float& w1, float& w2, float& w3)
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
__device__ __forceinline__ float bspline_fn(float t)
{
    t = fabsf(t);
    const float a = 2.0f - t;
    return (t < 1.0f) ? ((2.0f / 3.0f) - 0.5f * t * t * a) : ((t < 2.0f) ? (a * a * a * (1.0f / 6.0f)) : 0.0f);
}

template <bool SIMPLE>
__device__ __forceinline__ void cubic_weights(float f, float (&w)[4])
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
__device__ __forceinline__ int xcd_contiguous(int b, int n)
{
    const int xcd = b & 7, q = n >> 3, r = n & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (b >> 3);
}

template <int KIND /*0 linear, 1 cubic (bspline_weights), 2 cubic (bspline fn)*/, int TD, int TH, int TW, bool VEC4>
__global__ __launch_bounds__(256) void affine_tiled(const float* __restrict__ src, float* __restrict__ out,
                                                     const AffineParams p)
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
        any_valid = any_valid && (hi[r] >= p.vlo[r]) && (lo[r] < p.vhi[r]);
        all_valid = all_valid && (lo[r] >= p.vlo[r]) && (hi[r] < p.vhi[r]);
    }

    const int kw = tid % TW;
    const int jh0 = tid / TW;
    constexpr int RP = 256 / TW;          // tile rows covered per pass
    constexpr int NJ = TH / RP;
    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;

    if (!any_valid) {
        // the whole tile maps outside the valid interval: zero-fill (or leave untouched) and leave
        if (!keep) {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int h = h0 + jh0 + jj * RP, w = w0 + kw;
                if (h < p.oH && w < p.oW) {
                    for (int i = 0; i < TD; ++i) {
                        const int d = d0 + i;
                        if (d < p.oD) out[((int64_t)d * p.oH + h) * p.oW + w] = 0.0f;
                    }
                }
            }
        }
        return;
    }

    // integer origin of the staged box (the coordinates are finite and within +-2^30 here because the
    // tile intersects the valid interval and its extent was bounded on the host)
    int o[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = (int)floor(lo[r]) - HALO;
    if constexpr (VEC4) o[2] &= ~3;       // 16-byte aligned rows (W % 4 == 0 is a host-side precondition)

    // ---- stage the source box into LDS, zero outside the volume ----
    const int Lx = p.Lx, Ly = p.Ly, Lz = p.Lz;
    if constexpr (VEC4) {
        const int nvx = Lx >> 2;
        const int total = Lz * Ly * nvx;
        const int step_rows = 256 / nvx, step_cx = 256 - step_rows * nvx;
        const int step_z = step_rows / Ly, step_y = step_rows - step_z * Ly;
        int v = tid;
        int row = v / nvx;
        int cx = v - row * nvx;
        int z = row / Ly;
        int y = row - z * Ly;
        float4* lds4 = reinterpret_cast<float4*>(lds);
        for (; v < total; v += 256) {
            const int gz = o[0] + z, gy = o[1] + y, gx = o[2] + 4 * cx;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gz < (unsigned)p.sD && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sW)
                val = *reinterpret_cast<const float4*>(src + ((int64_t)gz * p.sH + gy) * p.sW + gx);
            lds4[v] = val;
            cx += step_cx; y += step_y; z += step_z;
            if (cx >= nvx) { cx -= nvx; y += 1; }
            if (y >= Ly) { y -= Ly; z += 1; }
        }
    } else {
        const int total = Lz * Ly * Lx;
        const int step_rows = 256 / Lx, step_cx = 256 - step_rows * Lx;
        const int step_z = step_rows / Ly, step_y = step_rows - step_z * Ly;
        int v = tid;
        int row = v / Lx;
        int cx = v - row * Lx;
        int z = row / Ly;
        int y = row - z * Ly;
        for (; v < total; v += 256) {
            const int gz = o[0] + z, gy = o[1] + y, gx = o[2] + cx;
            float val = 0.f;
            if ((unsigned)gz < (unsigned)p.sD && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sW)
                val = src[((int64_t)gz * p.sH + gy) * p.sW + gx];
            lds[v] = val;
            cx += step_cx; y += step_y; z += step_z;
            if (cx >= Lx) { cx -= Lx; y += 1; }
            if (y >= Ly) { y -= Ly; z += 1; }
        }
    }
    __syncthreads();

    // ---- gather ----
    // tile-relative coordinate of output voxel (i,j,k): rel_r = (base_r - o_r) + m[r][0]*i + m[r][1]*j + m[r][2]*k
    double b[3], vlo[3], vhi[3], cmax[3];
    const int Ldim[3] = {Lz, Ly, Lx};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        b[r] = base[r] - (double)o[r];
        vlo[r] = p.vlo[r] - (double)o[r];
        vhi[r] = p.vhi[r] - (double)o[r];
        // taps i-HALO .. i+1+HALO must lie in [0, L-1]  =>  rel in [HALO, L-1-HALO)
        cmax[r] = (double)(Ldim[r] - 1 - HALO) - 1e-9;
    }
    const int LyLx = Ly * Lx;

#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = jh0 + jj * RP;
        const int h = h0 + j, w = w0 + kw;
        if (h >= p.oH || w >= p.oW) continue;
        double c0 = fma(p.m[1], (double)j, fma(p.m[2], (double)kw, b[0]));
        double c1 = fma(p.m[5], (double)j, fma(p.m[6], (double)kw, b[1]));
        double c2 = fma(p.m[9], (double)j, fma(p.m[10], (double)kw, b[2]));
        float* optr = out + ((int64_t)d0 * p.oH + h) * p.oW + w;
        const int64_t ostride = (int64_t)p.oH * p.oW;
        const int nd = min(TD, p.oD - d0);
        for (int i = 0; i < nd; ++i, c0 += p.m[0], c1 += p.m[4], c2 += p.m[8], optr += ostride) {
            bool inside = true;
            if (!all_valid)
                inside = (c0 >= vlo[0]) && (c0 < vhi[0]) && (c1 >= vlo[1]) && (c1 < vhi[1]) && (c2 >= vlo[2]) && (c2 < vhi[2]);
            // clamp guards against float64 rounding differences between this path and the box origin
            const double z = fmin(fmax(c0, (double)HALO), cmax[0]);
            const double y = fmin(fmax(c1, (double)HALO), cmax[1]);
            const double x = fmin(fmax(c2, (double)HALO), cmax[2]);
            const double fzd = floor(z), fyd = floor(y), fxd = floor(x);
            const int iz = (int)fzd, iy = (int)fyd, ix = (int)fxd;
            const float fz = (float)(z - fzd), fy = (float)(y - fyd), fx = (float)(x - fxd);
            float val;
            if constexpr (!CUBIC) {
                const float* q = lds + (iz * Ly + iy) * Lx + ix;
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
                val = fmaf(fz, y1 - y0, y0);
            } else {
                float wx[4], wy[4], wz[4];
                cubic_weights<KIND == 2>(fx, wx);
                cubic_weights<KIND == 2>(fy, wy);
                cubic_weights<KIND == 2>(fz, wz);
                const float* q = lds + ((iz - 1) * Ly + (iy - 1)) * Lx + (ix - 1);
                val = 0.f;
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
            }
            if (inside) *optr = val;
            else if (!keep) *optr = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// direct kernel: one thread per output voxel, taps from global memory with explicit border tests
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fetch0(const float* __restrict__ src, const AffineParams& p, int z, int y, int x)
{
    if ((unsigned)z < (unsigned)p.sD && (unsigned)y < (unsigned)p.sH && (unsigned)x < (unsigned)p.sW)
        return src[((int64_t)z * p.sH + y) * p.sW + x];
    return 0.0f;
}

template <int KIND>
__global__ __launch_bounds__(256) void affine_direct(const float* __restrict__ src, float* __restrict__ out,
                                                      const AffineParams p)
{
    constexpr bool CUBIC = KIND != 0;
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
    float val;
    if constexpr (!CUBIC) {
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
        val = fmaf(fz, y1 - y0, y0);
    } else {
        float wx[4], wy[4], wz[4];
        cubic_weights<KIND == 2>(fx, wx);
        cubic_weights<KIND == 2>(fy, wy);
        cubic_weights<KIND == 2>(fz, wz);
        val = 0.f;
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
    }
    out[idx] = val;
}

// ---------------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------------
struct TileCfg { int td, th, tw; };
static const TileCfg kTiles[] = {
    {16, 16, 16},   // 0: cube -- smallest box for general 3-D rotations
    {8, 16, 32},    // 1: 128-byte store segments
    {8, 8, 32},     // 2: half-size for fat footprints
    {4, 8, 32},     // 3: minification up to ~3x
};

int tile_config_count() { return (int)(sizeof(kTiles) / sizeof(kTiles[0])); }
void tile_config(int idx, int* td, int* th, int* tw) { *td = kTiles[idx].td; *th = kTiles[idx].th; *tw = kTiles[idx].tw; }

typedef void (*tiled_fn)(const float*, float*, const AffineParams);

template <int TD, int TH, int TW>
static tiled_fn pick_tiled(int kind, bool vec4)
{
    switch (kind * 2 + (vec4 ? 1 : 0)) {
        case 0: return affine_tiled<0, TD, TH, TW, false>;
        case 1: return affine_tiled<0, TD, TH, TW, true>;
        case 2: return affine_tiled<1, TD, TH, TW, false>;
        case 3: return affine_tiled<1, TD, TH, TW, true>;
        case 4: return affine_tiled<2, TD, TH, TW, false>;
        default: return affine_tiled<2, TD, TH, TW, true>;
    }
}

static tiled_fn tiled_entry(int cfg, int kind, bool vec4)
{
    switch (cfg) {
        case 0: return pick_tiled<16, 16, 16>(kind, vec4);
        case 1: return pick_tiled<8, 16, 32>(kind, vec4);
        case 2: return pick_tiled<8, 8, 32>(kind, vec4);
        default: return pick_tiled<4, 8, 32>(kind, vec4);
    }
}

static int interp_kind(int interp)
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
            for (int v = 0; v < 2; ++v) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tiled_entry(cfg, kind, v != 0)),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

hipError_t launch_affine_tiled(int cfg, int interp, bool vec4, const float* src, float* out,
                               const AffineParams& p, int grid, int lds_bytes, hipStream_t stream)
{
    tiled_fn fn = tiled_entry(cfg, interp_kind(interp), vec4);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds_bytes, stream, src, out, p);
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

}  // namespace vt
