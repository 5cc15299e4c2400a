// vt_kernels_rows.hip -- kind 10: maps that leave axis 2 alone, src_w = w + t (any t since round 5: integer or fractional)
// ([a b 0 t0; c e 0 t1; 0 0 1 t]: every rotation about axis 2 through the default centre, scaled / sheared or not in the (d, h) plane).
//
// The resident plain layout [z][y][x] is, for these maps, what the plane-quad copy is for rotations about axis 0: the axis the map
// leaves alone is the CONTIGUOUS one.  A wave owns one output pixel (d, h) and 64 consecutive w: the in-plane stencil -- 2 x 2 or
// 4 x 4 source rows (z, y) -- and its weights are the same for all 64 lanes, every tap is one conflict-free `ds_read_b32` of 64
// consecutive floats of a staged source row, and the 64 results leave as one 256-byte store.  No exchanged copy, no exchanged
// result, no transpose pass (the marching kernels need all three for these maps: 0.21 + 0.19 ms at 512^3; writing the caller's
// array from a kernel that marches along w was measured in four forms and never beat that, profiles/r04_axis2_direct_output.txt).
//
// Arithmetic, operation for operation that of `affine_direct` / the oracle (transforms.py:253-281, helper_interpolation.h:3-68):
//   * coordinates of rows 0 and 1 through canonical_coord (their w term is fma(0, w, .) exactly), floor and float32 fraction;
//   * trilinear: the x-lerp of a row pair is fmaf(0, b - a, a) == a for finite data (the fraction along w is exactly 0), so one
//     tap per row; then the y- and z-lerps in the usual order;
//   * cubic: the x-sum of a row, w0 c[x-1] + w1 c[x] + w2 c[x+1] (+ 0 c[x+2]) with the weights of fraction 0, is formed ONCE, when the
//     x-convolved copy is built (relayout_xfir: the same three operations in the same order), and the kernel reads it as one tap;
//     then accy and the z sum as in direct_sample.  Bit-identical to affine_direct on finite data.
// A non-finite sample in the tap column of weight exactly 0 does not reach the output (as for KIND 3 / KIND 4).
#include "vt_internal.h"
#include "vt_device.h"

namespace vt {

constexpr int kRowPH = 8, kRowRun = 64;      // a wave's pixels (one d, eight h) and its run of w; a workgroup's tile is PD x 8 pixels, PD waves

// dst[z][y][x] = fmaf(w2, c[x+1], fmaf(w1, c[x], w0 * c[x-1])), zero border, pad columns zero
__global__ __launch_bounds__(256) void relayout_xfir(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int W, int P, int simple)
{
    float w[4];
    if (simple) cubic_weights<true>(0.0f, w); else cubic_weights<false>(0.0f, w);
    const int64_t row = (int64_t)blockIdx.y + (int64_t)gridDim.y * blockIdx.z;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows || x >= P) return;
    const float* s = src + row * P;
    float val = 0.0f;
    if (x < W) {
        const float cm = x > 0 ? s[x - 1] : 0.0f, c0 = s[x], cp = x + 1 < W ? s[x + 1] : 0.0f;
        val = fmaf(w[2], cp, fmaf(w[1], c0, w[0] * cm));
    }
    dst[row * P + x] = val;
}

hipError_t launch_relayout_xfir(const float* src, float* dst, int D, int H, int W, int P, bool simple, hipStream_t stream)
{
    const int64_t rows = (int64_t)D * H;
    const unsigned gy = (unsigned)std::min<int64_t>(rows, 65535), gz = (unsigned)((rows + gy - 1) / gy);
    if (rows <= 0 || gz > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(relayout_xfir, dim3((unsigned)((P + 255) / 256), gy, gz), dim3(256), 0, stream, src, dst, rows, W, P, simple ? 1 : 0);
    return hipGetLastError();
}

// KIND 0: trilinear, integer axis-2 offset (2 x 2 rows of the plain copy, one tap per row)
// KIND 1: cubic, integer offset (4 x 4 rows of the x-convolved copy, one tap per row; flag bit 18: `_simple` weights)
// KIND 2: trilinear, fractional offset (two taps per row, the x-lerp of direct_sample)
// KIND 3: cubic, fractional offset (4 x 4 rows of the PLAIN copy, four taps per row with the lane's own x weights)
// NV: 16-byte vectors staged per row -- 16 where the run starts on a vector of the source row (integer offset that is a multiple of four:
// the rotation about axis 2 through the default centre), 18 otherwise: the run starts up to three floats into its first vector and the
// fractional kinds read one (trilinear) or three (cubic) columns beyond the 64.  Any other offset used to take the axis exchange path
// (exchanged copy + plane-quad kernel + transpose pass, 0.43 ms at 512^3 against 0.31); the reference's kernel has no such cliff
// (transforms.py:269-274).
template <int KIND, int PD, int NV>
__global__ __launch_bounds__(64 * PD) void affine_rows(const float* __restrict__ src, float* __restrict__ out, const float* __restrict__ zeros16,
                                                   const AffineParams p)
{
    constexpr bool CUBIC = (KIND & 1) != 0, FRAC = KIND >= 2;
    constexpr int HALO = CUBIC ? 1 : 0, NT = CUBIC ? 4 : 2;
    constexpr int NX = !FRAC ? 1 : (CUBIC ? 4 : 2);            // taps per row along x
    constexpr int XH = (FRAC && CUBIC) ? 1 : 0;                // columns in front of floor(src_w) that a lane taps
    constexpr int RS = 4 * NV;                                 // staged floats per row
    constexpr unsigned RSB = 16u * NV;                         // ... in bytes
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int run = blockIdx.x, th_i = blockIdx.y, td_i = blockIdx.z;
    constexpr int kRowPD = PD, NTHR = 64 * PD;
    const int d0 = td_i * kRowPD, h0 = th_i * kRowPH, w0 = run * kRowRun;
    const int Ly = p.Ly, Lz = p.Lz;
    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;

    // (z, y) box of the tile's taps: rows 0 and 1 of the matrix ignore w
    double base[2], lo[2], hi[2];
    bool any_valid = true;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        base[r] = fma(p.m[4 * r], (double)d0, fma(p.m[4 * r + 1], (double)h0, p.m[4 * r + 3]));
        lo[r] = base[r] + p.neg[r];
        hi[r] = base[r] + p.pos[r];
        any_valid = any_valid && (hi[r] >= p.vlo[r] - kTileMargin) && (lo[r] < p.vhi[r] + kTileMargin);
    }
    const int t2 = p.zoff;                        // floor of the axis-2 offset: src_w = w + t2 (+ a fraction for the kinds 2 / 3)
    any_valid = any_valid && ((double)(w0 + kRowRun + t2) >= p.vlo[2] - kTileMargin) && ((double)(w0 + t2) < p.vhi[2] + kTileMargin);
    const int w = w0 + lane;
    if (!any_valid) {
        if (!keep && w < p.oW) {
            const int d = d0 + wv;
            if (d < p.oD)
                for (int i = 0; i < kRowPH && h0 + i < p.oH; ++i) out[((int64_t)d * p.oH + (h0 + i)) * p.oW + w] = 0.0f;
        }
        return;
    }
    int oz, oy;
    {
        // The box origin lies 1e-9 below the tile's lowest coordinate (and the planner sizes the box for an extent 1e-8 larger): a pixel's
        // own fma chain may round across an integer the tile corners' chain did not -- by 1e-13 at most --, and its taps must still index
        // inside the box.  (Until round 5 such a pixel would have read its taps from global memory: a path no test ever reached.)
        const int f0 = (int)floor(lo[0] - 1.0e-9), f1 = (int)floor(lo[1] - 1.0e-9);
        // (the builtin, not an inline-asm v_readfirstlane: the compiler's hazard recogniser does not look into asm statements -- no wait
        //  state between the v_cvt_i32_f64 that produces the value and the lane read of it --, and the cubic instantiation got a stale
        //  origin in some waves: rows staged by different waves then disagreed and results changed from launch to launch)
        oz = __builtin_amdgcn_readfirstlane(f0) - HALO;
        oy = __builtin_amdgcn_readfirstlane(f1) - HALO;
    }
    // first staged column: the vector that holds the run's first tap (a multiple of four; for NV = 16 that tap itself)
    const int x0 = (w0 + t2 - XH) & ~3;

    // ---- stage Lz x Ly source rows, NV vectors of 16 bytes each: vector v lands at lds + 16 v ----
    {
        const int total = Lz * Ly * NV;
        const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
        for (int vb = wave_first; vb < total; vb += NTHR) {
            const int v = vb + lane;
            const int row = NV == 16 ? (v >> 4) : (int)__umulhi((unsigned)v, 0x0E38E38Fu);      // v / 18 for v < 2^24 (ceil(2^32 / 18))
            const int seg = v - row * NV;
            const int zz = (int)__umulhi((unsigned)row, p.psv_magic);        // row / Ly (host constant: floor(2^32 / Ly) + 1)
            const int yy = row - zz * Ly;
            const int gz = oz + zz, gy = oy + yy, gx = x0 + 4 * seg;
            const bool inb = (unsigned)gz < (unsigned)p.sD && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sP;
            const float* g = inb ? src + (((int64_t)gz * p.sH + gy) * p.sP + gx) : zeros16;
            if (v < total)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lds + 4 * vb), 16, 0, 0);
        }
    }

    // ---- one wave per d, eight pixels (h) each ----
    // Everything about a pixel but its 64 samples is the same for the wave's 64 lanes: lane i < 8 works out pixel i (coordinates in
    // float64, fractions, the eight weights, the first tap row) while the staged rows are still in flight -- as vector work per pixel
    // the same arithmetic was ~150 instructions for 64 voxels, three times the taps and sums ([measured] 512^3 cubic 0.66 ms in that form) --
    const int d = d0 + wv;
    const int hl = h0 + (lane & 7);
    const double s0 = canonical_coord(p, 0, d, hl, 0), s1 = canonical_coord(p, 1, d, hl, 0);      // the w column of rows 0 / 1 is exactly 0
    const double fzd = floor(s0), fyd = floor(s1);
    const float fz = (float)(s0 - fzd), fy = (float)(s1 - fyd);
    float wy[4] = {0.f, 0.f, 0.f, 0.f}, wz[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CUBIC) {
        if (p.flags & (1 << 18)) { cubic_weights<true>(fy, wy); cubic_weights<true>(fz, wz); }
        else { cubic_weights<false>(fy, wy); cubic_weights<false>(fz, wz); }
    }
    const int rz_l = (int)fzd - HALO - oz, ry_l = (int)fyd - HALO - oy;       // inside the box by construction (see the origin above)
    const int in_zy_l = ((s0 >= p.vlo[0]) && (s0 < p.vhi[0]) && (s1 >= p.vlo[1]) && (s1 < p.vhi[1]) && hl < p.oH && d < p.oD) ? 1 : 0;
    // ... and leaves them in LDS, 16 dwords per pixel behind the staged rows (every lane of the wave reads them back from one address: a
    // broadcast).  (Taking them across lanes with v_readlane instead gave results that changed from launch to launch.)
    float* const prm = lds + Lz * Ly * RS + (wv * kRowPH) * 16;
    if (lane < kRowPH) {
        float* q = prm + lane * 16;
        typedef float v4f __attribute__((ext_vector_type(4)));
        *reinterpret_cast<v4f*>(q) = v4f{wy[0], wy[1], wy[2], wy[3]};
        *reinterpret_cast<v4f*>(q + 4) = v4f{wz[0], wz[1], wz[2], wz[3]};
        typedef int v4i __attribute__((ext_vector_type(4)));
        // (integers travel as integers: a small integer's bit pattern is a subnormal float, and float moves may flush it)
        *reinterpret_cast<v4f*>(q + 8) = v4f{fy, fz, 0.f, 0.f};
        *reinterpret_cast<v4i*>(q + 12) = v4i{rz_l * Ly + ry_l, in_zy_l, rz_l, ry_l};
    }

    // the lane's own column: src_w by the canonical chain of row 2 (fma(0, d, fma(0, h, fma(1, w, t)))), its floor and float32 fraction,
    // the x weights of a fractional cubic offset -- exactly what affine_direct forms per voxel
    const double sw = (double)w + p.m[11];
    const bool in_x = (sw >= p.vlo[2]) && (sw < p.vhi[2]);
    const double fxd = floor(sw);
    const float fx = (float)(sw - fxd);
    float wx[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FRAC && CUBIC) {
        if (p.flags & (1 << 18)) cubic_weights<true>(fx, wx); else cubic_weights<false>(fx, wx);
    }
    int col = (int)fxd - XH - x0;                                              // 0 .. 3 + lane for every lane whose column matters
    col = min(max(col, 0), RS - NX);                                           // (lanes beyond the output's width: any column of the row)

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (d >= p.oD) return;
    const unsigned lds_b = lds_byte_address(lds) + 4u * (unsigned)col;
    float* optr = out + ((int64_t)d * p.oH + h0) * p.oW + w;
#pragma unroll
    for (int i = 0; i < kRowPH; ++i) {
        if (h0 + i >= p.oH) break;
        // every branch up to the store is uniform over the wave: the samples of all 64 lanes are formed (columns outside the volume were
        // staged as zeros), the store alone looks at the lane's own w
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f pwy = *reinterpret_cast<const v4f*>(prm + i * 16), pwz = *reinterpret_cast<const v4f*>(prm + i * 16 + 4);
        const v4f pf = *reinterpret_cast<const v4f*>(prm + i * 16 + 8);
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i pi = *reinterpret_cast<const v4i*>(prm + i * 16 + 12);
        const bool inside_zy = (__builtin_amdgcn_readfirstlane(pi[1]) & 1) != 0;
        float val = 0.f;
        if (inside_zy) {
            const unsigned a0 = lds_b + RSB * (unsigned)__builtin_amdgcn_readfirstlane(pi[0]);
            float t[NT][NT];
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const unsigned ra = a0 + RSB * (unsigned)(c * Ly + b);
                    auto tap = [&](int k) { return *reinterpret_cast<const __attribute__((address_space(3))) float*>((size_t)(ra + 4u * (unsigned)k)); };
                    if constexpr (NX == 1) {
                        t[c][b] = tap(0);
                    } else if constexpr (NX == 2) {
                        const float a = tap(0), bb = tap(1);
                        t[c][b] = fmaf(fx, bb - a, a);                         // direct_sample's x-lerp
                    } else {
                        float accx = wx[0] * tap(0);                           // direct_sample's x-sum
                        accx = fmaf(wx[1], tap(1), accx);
                        accx = fmaf(wx[2], tap(2), accx);
                        t[c][b] = fmaf(wx[3], tap(3), accx);
                    }
                }
            if constexpr (!CUBIC) {
                const float y0 = fmaf(pf[0], t[0][1] - t[0][0], t[0][0]);
                const float y1 = fmaf(pf[0], t[1][1] - t[1][0], t[1][0]);
                val = fmaf(pf[1], y1 - y0, y0);
            } else {
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    float accy = 0.f;
#pragma unroll
                    for (int b = 0; b < NT; ++b) accy = fmaf(pwy[b], t[c][b], accy);
                    val = fmaf(pwz[c], accy, val);
                }
            }
        }
        if (w < p.oW) {
            if (inside_zy && in_x) optr[(int64_t)i * p.oW] = val;
            else if (!keep) optr[(int64_t)i * p.oW] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same tile walking ALL its runs of 64 w with two row buffers (round 5): the box of source rows (z, y) a pixel tile taps does not
// depend on w, so the tile's geometry and its pixels' parameters are worked out once, and run k+1 is staged while run k is computed.
// [counters of the one-run-per-workgroup form, profiles/r05_axis2_512_*_summary.json] HBM-side traffic 0.99-1.00 GB (0.93x algorithmic: only
// the part of the source the output maps to is read), no LDS conflicts, but SQ_WAIT_ANY is 50 % of the wave cycles: stage -> wait for every
// byte -> one barrier -> compute, nothing overlapping inside a workgroup.  Here: one barrier per run, a counted `s_waitcnt vmcnt(8)` -- every
// pixel issues exactly one store per wave, lanes that must not write carry an offset beyond the output descriptor's records, so the eight
// stores of a run are always eight instructions behind the next run's staging loads and may stay in flight while those are waited for.
// ---------------------------------------------------------------------------------------------------
template <int KIND, int PD, int NV>
__global__ __launch_bounds__(64 * PD) void affine_rows_db(const float* __restrict__ src, float* __restrict__ out, const float* __restrict__ zeros16,
                                                      const AffineParams p)
{
    constexpr bool CUBIC = (KIND & 1) != 0, FRAC = KIND >= 2;
    constexpr int HALO = CUBIC ? 1 : 0, NT = CUBIC ? 4 : 2;
    constexpr int NX = !FRAC ? 1 : (CUBIC ? 4 : 2);
    constexpr int XH = (FRAC && CUBIC) ? 1 : 0;
    constexpr int RS = 4 * NV;
    constexpr unsigned RSB = 16u * NV;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int th_i = blockIdx.x, td_i = blockIdx.y;
    constexpr int NTHR = 64 * PD;
    const int d0 = td_i * PD, h0 = th_i * kRowPH;
    const int Ly = p.Ly, Lz = p.Lz;
    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;
    const int nruns = (p.oW + kRowRun - 1) / kRowRun;
    const int d = d0 + wv;

    double base[2], lo[2], hi[2];
    bool any_valid = true;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        base[r] = fma(p.m[4 * r], (double)d0, fma(p.m[4 * r + 1], (double)h0, p.m[4 * r + 3]));
        lo[r] = base[r] + p.neg[r];
        hi[r] = base[r] + p.pos[r];
        any_valid = any_valid && (hi[r] >= p.vlo[r] - kTileMargin) && (lo[r] < p.vhi[r] + kTileMargin);
    }
    if (!any_valid) {
        // no pixel of the tile maps inside the valid (z, y) interval, whatever w: zero-fill its rows (or leave them untouched)
        if (!keep && d < p.oD)
            for (int i = 0; i < kRowPH && h0 + i < p.oH; ++i) {
                float* row = out + ((int64_t)d * p.oH + (h0 + i)) * p.oW;
                for (int w = lane; w < p.oW; w += 64) row[w] = 0.0f;
            }
        return;
    }
    int oz, oy;
    {
        const int f0 = (int)floor(lo[0] - 1.0e-9), f1 = (int)floor(lo[1] - 1.0e-9);       // (see affine_rows)
        oz = __builtin_amdgcn_readfirstlane(f0) - HALO;
        oy = __builtin_amdgcn_readfirstlane(f1) - HALO;
    }
    const int t2 = p.zoff;
    const int box_floats = Lz * Ly * RS;
    float* const prm = lds + 2 * box_floats + (wv * kRowPH) * 16;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int total = Lz * Ly * NV;

    // stage run r into buffer r & 1: NV vectors per row from column x0(r) on; vectors outside the volume come from a block of zeros
    auto stage = [&](int r) {
        const int x0 = (r * kRowRun + t2 - XH) & ~3;
        float* const dstb = lds + (r & 1) * box_floats;
        for (int vb = wave_first; vb < total; vb += NTHR) {
            const int v = vb + lane;
            const int row = NV == 16 ? (v >> 4) : (int)__umulhi((unsigned)v, 0x0E38E38Fu);
            const int seg = v - row * NV;
            const int zz = (int)__umulhi((unsigned)row, p.psv_magic);
            const int yy = row - zz * Ly;
            const int gz = oz + zz, gy = oy + yy, gx = x0 + 4 * seg;
            const bool inb = (unsigned)gz < (unsigned)p.sD && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sP;
            const float* g = inb ? src + (((int64_t)gz * p.sH + gy) * p.sP + gx) : zeros16;
            if (v < total)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(dstb + 4 * vb), 16, 0, 0);
        }
    };
    stage(0);

    // the pixels' parameters, once per tile (lane i < 8 of wave d works out pixel (d, h0 + i)), left in LDS behind the two buffers
    {
        const int hl = h0 + (lane & 7);
        const double s0 = canonical_coord(p, 0, d, hl, 0), s1 = canonical_coord(p, 1, d, hl, 0);
        const double fzd = floor(s0), fyd = floor(s1);
        const float fz = (float)(s0 - fzd), fy = (float)(s1 - fyd);
        float wy[4] = {0.f, 0.f, 0.f, 0.f}, wz[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (CUBIC) {
            if (p.flags & (1 << 18)) { cubic_weights<true>(fy, wy); cubic_weights<true>(fz, wz); }
            else { cubic_weights<false>(fy, wy); cubic_weights<false>(fz, wz); }
        }
        const int rz_l = (int)fzd - HALO - oz, ry_l = (int)fyd - HALO - oy;
        const int in_zy_l = ((s0 >= p.vlo[0]) && (s0 < p.vhi[0]) && (s1 >= p.vlo[1]) && (s1 < p.vhi[1]) && hl < p.oH && d < p.oD) ? 1 : 0;
        if (lane < kRowPH) {
            float* q = prm + lane * 16;
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef int v4i __attribute__((ext_vector_type(4)));
            *reinterpret_cast<v4f*>(q) = v4f{wy[0], wy[1], wy[2], wy[3]};
            *reinterpret_cast<v4f*>(q + 4) = v4f{wz[0], wz[1], wz[2], wz[3]};
            *reinterpret_cast<v4f*>(q + 8) = v4f{fy, fz, 0.f, 0.f};
            *reinterpret_cast<v4i*>(q + 12) = v4i{rz_l * Ly + ry_l, in_zy_l | ((hl < p.oH && d < p.oD) ? 2 : 0), rz_l, ry_l};
        }
    }
    // the output rows of this wave's pixels through one descriptor: a lane that must not write takes an offset beyond its records
    const int64_t orow0 = ((int64_t)min(d, p.oD - 1) * p.oH + h0) * p.oW;
    const int orow_b = p.oW * 4;
    __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out + orow0), 0, (unsigned)min((int64_t)kRowPH * orow_b, (int64_t)0x7fffffff), 0x00020000);
    constexpr int kDrop = 0x7ffffff0;             // beyond every descriptor's records: the store is dropped

    bool first = true;
    for (int r = 0; r < nruns; ++r) {
        // run r has landed (this wave's part); the eight stores of run r - 1 were issued behind its loads and may stay in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(kRowPH) : "memory");
        first = false;
        __syncthreads();                          // ... everywhere; everyone has finished reading buffer (r + 1) & 1
        if (r + 1 < nruns) stage(r + 1);
        const int w = r * kRowRun + lane;
        const double sw = (double)w + p.m[11];
        const bool in_x = (sw >= p.vlo[2]) && (sw < p.vhi[2]);
        const double fxd = floor(sw);
        const float fx = (float)(sw - fxd);
        float wx[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FRAC && CUBIC) {
            if (p.flags & (1 << 18)) cubic_weights<true>(fx, wx); else cubic_weights<false>(fx, wx);
        }
        const int x0 = (r * kRowRun + t2 - XH) & ~3;
        int col = (int)fxd - XH - x0;
        col = min(max(col, 0), RS - NX);
        const unsigned lds_b = lds_byte_address(lds + (r & 1) * box_floats) + 4u * (unsigned)col;
        const int wofs = (w < p.oW) ? 4 * w : kDrop;
#pragma unroll
        for (int i = 0; i < kRowPH; ++i) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4f pwy = *reinterpret_cast<const v4f*>(prm + i * 16), pwz = *reinterpret_cast<const v4f*>(prm + i * 16 + 4);
            const v4f pf = *reinterpret_cast<const v4f*>(prm + i * 16 + 8);
            const v4i pi = *reinterpret_cast<const v4i*>(prm + i * 16 + 12);
            const int fl = __builtin_amdgcn_readfirstlane(pi[1]);
            const bool inside_zy = (fl & 1) != 0, exists = (fl & 2) != 0;
            float val = 0.f;
            if (inside_zy) {
                const unsigned a0 = lds_b + RSB * (unsigned)__builtin_amdgcn_readfirstlane(pi[0]);
                float t[NT][NT];
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int b = 0; b < NT; ++b) {
                        const unsigned ra = a0 + RSB * (unsigned)(c * Ly + b);
                        auto tap = [&](int k) { return *reinterpret_cast<const __attribute__((address_space(3))) float*>((size_t)(ra + 4u * (unsigned)k)); };
                        if constexpr (NX == 1) {
                            t[c][b] = tap(0);
                        } else if constexpr (NX == 2) {
                            const float a = tap(0), bb = tap(1);
                            t[c][b] = fmaf(fx, bb - a, a);
                        } else {
                            float accx = wx[0] * tap(0);
                            accx = fmaf(wx[1], tap(1), accx);
                            accx = fmaf(wx[2], tap(2), accx);
                            t[c][b] = fmaf(wx[3], tap(3), accx);
                        }
                    }
                if constexpr (!CUBIC) {
                    const float y0 = fmaf(pf[0], t[0][1] - t[0][0], t[0][0]);
                    const float y1 = fmaf(pf[0], t[1][1] - t[1][0], t[1][0]);
                    val = fmaf(pf[1], y1 - y0, y0);
                } else {
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        float accy = 0.f;
#pragma unroll
                        for (int b = 0; b < NT; ++b) accy = fmaf(pwy[b], t[c][b], accy);
                        val = fmaf(pwz[c], accy, val);
                    }
                }
            }
            // exactly one store instruction per pixel and wave: the value, a zero, or -- where nothing may be written (a pixel or a column
            // beyond the output, an outside voxel under keep_outside) -- a store the descriptor drops
            const bool write_val = inside_zy && in_x;
            const int ofs = (exists && (write_val || !keep)) ? wofs : kDrop;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, write_val ? val : 0.0f), orsrc, ofs, i * orow_b, 0);
        }
    }
}

typedef void (*rows_fn)(const float*, float*, const float*, const AffineParams);
template <int PD>
static rows_fn rows_entry_pd(int kind, int nv)
{
    if (nv == 16) return kind == 0 ? affine_rows<0, PD, 16> : affine_rows<1, PD, 16>;
    switch (kind) {
        case 0: return affine_rows<0, PD, 18>;
        case 1: return affine_rows<1, PD, 18>;
        case 2: return affine_rows<2, PD, 18>;
        default: return affine_rows<3, PD, 18>;
    }
}
static rows_fn rows_entry(int kind, int pd, int nv) { return pd == 8 ? rows_entry_pd<8>(kind, nv) : rows_entry_pd<4>(kind, nv); }
template <int PD>
static rows_fn rows_db_entry_pd(int kind, int nv)
{
    if (nv == 16) return kind == 0 ? affine_rows_db<0, PD, 16> : affine_rows_db<1, PD, 16>;
    switch (kind) {
        case 0: return affine_rows_db<0, PD, 18>;
        case 1: return affine_rows_db<1, PD, 18>;
        case 2: return affine_rows_db<2, PD, 18>;
        default: return affine_rows_db<3, PD, 18>;
    }
}
static rows_fn rows_db_entry(int kind, int pd, int nv) { return pd == 8 ? rows_db_entry_pd<8>(kind, nv) : rows_db_entry_pd<4>(kind, nv); }

hipError_t init_rows_kernels()
{
    for (int kind = 0; kind < 4; ++kind)
        for (int pd = 4; pd <= 8; pd += 4)
            for (int nv = 16; nv <= 18; nv += 2) {
                if (nv == 16 && kind >= 2) continue;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rows_entry(kind, pd, nv)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(rows_db_entry(kind, pd, nv)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

void rows_tile(int* ph, int* run) { *ph = kRowPH; *run = kRowRun; }

// p.Lx: staged floats per row (64 or 72); flag bit 16: the axis-2 offset has a fraction (kinds 2 / 3)
hipError_t launch_affine_rows(int interp, int pd, const float* src, float* out, const float* zeros16, const AffineParams& p, int lds_bytes, hipStream_t stream)
{
    const int kind = (interp_kind(interp) == 0 ? 0 : 1) + ((p.flags & (1 << 16)) ? 2 : 0);
    if (p.flags & (1 << 17)) {                    // two row buffers, a workgroup walks all runs of its pixel tile (plan_rows)
        const dim3 g2((unsigned)((p.oH + kRowPH - 1) / kRowPH), (unsigned)((p.oD + pd - 1) / pd));
        hipLaunchKernelGGL(rows_db_entry(kind, pd, p.Lx == 64 ? 16 : 18), g2, dim3(64 * pd), lds_bytes, stream, src, out, zeros16, p);
        return hipGetLastError();
    }
    const dim3 g((unsigned)((p.oW + kRowRun - 1) / kRowRun), (unsigned)((p.oH + kRowPH - 1) / kRowPH), (unsigned)((p.oD + pd - 1) / pd));
    hipLaunchKernelGGL(rows_entry(kind, pd, p.Lx == 64 ? 16 : 18), g, dim3(64 * pd), lds_bytes, stream, src, out, zeros16, p);
    return hipGetLastError();
}

}  // namespace vt
