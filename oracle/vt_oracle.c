/*
 * vt_oracle.c -- CPU restatement of the reference's GPU hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP kernels: a plain-C restatement of what the
 * reference (the-lay/voltools v0.6.0, /root/reference) computes on its `device='gpu'` path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing in
 * voltools_amd/ imports, links or calls it, and the product path has no CPU fallback.
 *
 * Pinning.  The reference ships no golden vectors or assertions for this path:
 * its tests are empty stubs.  The oracle is therefore pinned by (a) outputs of the reference's own
 * CPU device path (`voltools.affine(..., device='cpu')`, i.e. scipy.ndimage.affine_transform with the
 * arguments of transforms.py:126-152) generated in the build container by tests/golden/make_golden.py
 * and committed under tests/golden/, compared on the interior mask where the two boundary contracts
 * agree, and (b) analytic known answers (identity, integer shifts, constants,
 * impulse response, partition of unity).  The reference's CUDA sources cannot be compiled here
 * (helper_math.h needs CUDA's vector types and texture intrinsics; no stand-ins are written), so there
 * is no oracle/_ref build.
 *
 * What is restated (file:line into /root/reference/voltools):
 *   transform kernel body ........ transforms.py:253-281  (index -> (d,h,w), M.(d,h,w,1)+0.5, skirt test)
 *   texture fetch ................ transforms.py:184-192  (unnormalised coords, linear filter, border = 0)
 *   linearTex3D .................. kernels/helper_interpolation.h:3-6
 *   cubicTex3D ................... kernels/helper_interpolation.h:8-40  (8 trilinear fetches at h0/h1)
 *   cubicTex3DSimple ............. kernels/helper_interpolation.h:42-68 (64 point fetches, bspline(t))
 *   bspline_weights, bspline ..... kernels/bspline.h:102-122
 *   prefilter (causal/anticausal). kernels/bspline.h:2-54, X/Y/Z passes :58-99, launcher transforms.py:290-309
 *
 * Two coordinate modes:
 *   default ................ float64 coordinate arithmetic from the float32 matrix entries (what the
 *                            HIP kernels do; removes the N-dependent float32 coordinate error), then
 *                            float32 weights and sums exactly as the reference's device functions.
 *   VT_ORACLE_FAITHFUL ..... float32 coordinate arithmetic exactly as written in transforms.py:265-274
 *                            (`dot(voxf, xform[r]) + .5f`) and the 8-fetch formulation of cubicTex3D.
 * The texture unit's 8-bit fixed-point interpolation weights (CUDA hardware detail) are restated
 * only when VT_ORACLE_TEXFRAC8 is set; the HIP kernels deliberately do not reproduce that error.
 *
 * Arithmetic is plain IEEE float32 without contraction (build with -ffp-contract=off).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VT_ORACLE_FAITHFUL     1   /* float32 coordinates + as-written cubicTex3D formulation */
#define VT_ORACLE_KEEP_OUTSIDE 2   /* `continue` for outside voxels (transforms.py:278) instead of writing 0 */
#define VT_ORACLE_TEXFRAC8     4   /* quantise trilinear fractions to 8 bits like the CUDA texture unit */

enum { LINEAR = 0, BSPLINE = 1, BSPLINE_SIMPLE = 2, FILT_BSPLINE = 3, FILT_BSPLINE_SIMPLE = 4 };

typedef struct {
    const float* v;
    int64_t D, H, W;        /* storage dims */
    int texfrac8;
} tex_t;

/* ---- texture fetches (transforms.py:184-192): border address mode, point / linear filter ---- */

/* tex3D with cudaFilterModePoint semantics at a texel centre is never used by the reference; the
 * "point fetches" of cubicTex3DSimple are linear-filter fetches at exact texel centres (u = i+0.5),
 * where the linear filter degenerates to the single texel i.  texel() is that texel with border 0. */
static inline float texel(const tex_t* t, int64_t z, int64_t y, int64_t x)
{
    if (z < 0 || y < 0 || x < 0 || z >= t->D || y >= t->H || x >= t->W) return 0.0f;
    return t->v[(z * t->H + y) * t->W + x];
}

static inline float q8(float a, int on)
{
    /* CUDA programming guide, "Linear Filtering": the fractional part is stored in 9-bit fixed point
     * with 8 bits of fractional value. */
    return on ? floorf(a * 256.0f + 0.5f) / 256.0f : a;
}

/* tex3D<float>(tex, x, y, z), linear filter, unnormalised coordinates: xB = x - 0.5, i = floor(xB),
 * a = frac(xB); result = sum over the 8 neighbours with weights (1-a)/a. (x is the fastest axis.) */
static float tex3d_linear(const tex_t* t, float x, float y, float z)
{
    float xb = x - 0.5f, yb = y - 0.5f, zb = z - 0.5f;
    float fx = floorf(xb), fy = floorf(yb), fz = floorf(zb);
    float a = q8(xb - fx, t->texfrac8), b = q8(yb - fy, t->texfrac8), c = q8(zb - fz, t->texfrac8);
    int64_t i = (int64_t)fx, j = (int64_t)fy, k = (int64_t)fz;
    float r = 0.0f;
    r += (1 - a) * (1 - b) * (1 - c) * texel(t, k, j, i);
    r += a * (1 - b) * (1 - c) * texel(t, k, j, i + 1);
    r += (1 - a) * b * (1 - c) * texel(t, k, j + 1, i);
    r += a * b * (1 - c) * texel(t, k, j + 1, i + 1);
    r += (1 - a) * (1 - b) * c * texel(t, k + 1, j, i);
    r += a * (1 - b) * c * texel(t, k + 1, j, i + 1);
    r += (1 - a) * b * c * texel(t, k + 1, j + 1, i);
    r += a * b * c * texel(t, k + 1, j + 1, i + 1);
    return r;
}

/* ---- bspline.h:102-122 ---- */
static inline void bspline_weights(float f, float* w0, float* w1, float* w2, float* w3)
{
    const float one_frac = 1.0f - f;
    const float squared = f * f;
    const float one_sqd = one_frac * one_frac;
    *w0 = 1.0f / 6.0f * one_sqd * one_frac;
    *w1 = 2.0f / 3.0f - 0.5f * squared * (2.0f - f);
    *w2 = 2.0f / 3.0f - 0.5f * one_sqd * (2.0f - one_frac);
    *w3 = 1.0f / 6.0f * squared * f;
}

static inline float bspline(float t)
{
    t = fabsf(t);
    const float a = 2.0f - t;
    if (t < 1.0f) return 2.0f / 3.0f - 0.5f * t * t * a;
    else if (t < 2.0f) return a * a * a / 6.0f;
    else return 0.0f;
}

/* ---- helper_interpolation.h:8-40, as written: 8 linear fetches at h0/h1 (coord = src + 0.5) ---- */
static float cubic_tex3d_asis(const tex_t* t, float cx, float cy, float cz)
{
    const float gx = cx - 0.5f, gy = cy - 0.5f, gz = cz - 0.5f;
    const float ix = floorf(gx), iy = floorf(gy), iz = floorf(gz);
    float w0x, w1x, w2x, w3x, w0y, w1y, w2y, w3y, w0z, w1z, w2z, w3z;
    bspline_weights(gx - ix, &w0x, &w1x, &w2x, &w3x);
    bspline_weights(gy - iy, &w0y, &w1y, &w2y, &w3y);
    bspline_weights(gz - iz, &w0z, &w1z, &w2z, &w3z);
    const float g0x = w0x + w1x, g1x = w2x + w3x, h0x = (w1x / g0x) - 0.5f + ix, h1x = (w3x / g1x) + 1.5f + ix;
    const float g0y = w0y + w1y, g1y = w2y + w3y, h0y = (w1y / g0y) - 0.5f + iy, h1y = (w3y / g1y) + 1.5f + iy;
    const float g0z = w0z + w1z, g1z = w2z + w3z, h0z = (w1z / g0z) - 0.5f + iz, h1z = (w3z / g1z) + 1.5f + iz;

    float tex000 = tex3d_linear(t, h0x, h0y, h0z);
    float tex100 = tex3d_linear(t, h1x, h0y, h0z);
    tex000 = g0x * tex000 + g1x * tex100;
    float tex010 = tex3d_linear(t, h0x, h1y, h0z);
    float tex110 = tex3d_linear(t, h1x, h1y, h0z);
    tex010 = g0x * tex010 + g1x * tex110;
    tex000 = g0y * tex000 + g1y * tex010;
    float tex001 = tex3d_linear(t, h0x, h0y, h1z);
    float tex101 = tex3d_linear(t, h1x, h0y, h1z);
    tex001 = g0x * tex001 + g1x * tex101;
    float tex011 = tex3d_linear(t, h0x, h1y, h1z);
    float tex111 = tex3d_linear(t, h1x, h1y, h1z);
    tex011 = g0x * tex011 + g1x * tex111;
    tex001 = g0y * tex001 + g1y * tex011;
    return g0z * tex000 + g1z * tex001;
}

/* Same sum with the fractions handed in (float64-coordinate mode): 4x4x4 taps at i-1..i+2 with the
 * bspline_weights() weights -- what the 8 trilinear fetches of cubicTex3D add up to. */
static float cubic_taps(const tex_t* t, int64_t iz, int64_t iy, int64_t ix, float fz, float fy, float fx, int simple)
{
    float wx[4], wy[4], wz[4];
    if (!simple) {
        bspline_weights(fx, &wx[0], &wx[1], &wx[2], &wx[3]);
        bspline_weights(fy, &wy[0], &wy[1], &wy[2], &wy[3]);
        bspline_weights(fz, &wz[0], &wz[1], &wz[2], &wz[3]);
        float acc = 0.0f;
        for (int c = 0; c < 4; ++c) {
            float accy = 0.0f;
            for (int b = 0; b < 4; ++b) {
                float accx = 0.0f;
                for (int a = 0; a < 4; ++a) accx += wx[a] * texel(t, iz - 1 + c, iy - 1 + b, ix - 1 + a);
                accy += wy[b] * accx;
            }
            acc += wz[c] * accy;
        }
        return acc;
    }
    /* helper_interpolation.h:42-68: weights bspline(offset - fraction), product of three, running sum */
    float result = 0.0f;
    for (int c = -1; c <= 2; ++c) {
        float bz = bspline((float)c - fz);
        for (int b = -1; b <= 2; ++b) {
            float byz = bspline((float)b - fy) * bz;
            for (int a = -1; a <= 2; ++a) {
                float bxyz = bspline((float)a - fx) * byz;
                result += bxyz * texel(t, iz + c, iy + b, ix + a);
            }
        }
    }
    return result;
}

static float linear_taps(const tex_t* t, int64_t iz, int64_t iy, int64_t ix, float fz, float fy, float fx)
{
    float a = q8(fx, t->texfrac8), b = q8(fy, t->texfrac8), c = q8(fz, t->texfrac8);
    float r = 0.0f;
    r += (1 - a) * (1 - b) * (1 - c) * texel(t, iz, iy, ix);
    r += a * (1 - b) * (1 - c) * texel(t, iz, iy, ix + 1);
    r += (1 - a) * b * (1 - c) * texel(t, iz, iy + 1, ix);
    r += a * b * (1 - c) * texel(t, iz, iy + 1, ix + 1);
    r += (1 - a) * (1 - b) * c * texel(t, iz + 1, iy, ix);
    r += a * (1 - b) * c * texel(t, iz + 1, iy, ix + 1);
    r += (1 - a) * b * c * texel(t, iz + 1, iy + 1, ix);
    r += a * b * c * texel(t, iz + 1, iy + 1, ix + 1);
    return r;
}

/*
 * The transform kernel (transforms.py:253-281) generalised the way the C ABI needs it:
 *   src ......... sD x sH x sW float32 (storage), whose plane 0 is plane `plane0` of a global volume
 *                 with gD planes (plane0 = 0, gD = sD for an ordinary volume); planes outside the
 *                 storage window read as 0.
 *   out ......... oD x oH x oW float32; output voxel (d,h,w) is global output voxel (d + out_plane0, h, w).
 *   m ........... 16 doubles, row-major 4x4 pull matrix (rows 0..2 used).  The float32 matrix of the
 *                 reference is passed here converted to double (exact).
 */
int vt_oracle_affine_ex(const float* src, int64_t sD, int64_t sH, int64_t sW,
                        int64_t plane0, int64_t gD,
                        float* out, int64_t oD, int64_t oH, int64_t oW, int64_t out_plane0,
                        const double* m, int interp, int flags)
{
    if (!src || !out || !m || sD <= 0 || sH <= 0 || sW <= 0 || oD <= 0 || oH <= 0 || oW <= 0) return 1;
    if (interp < LINEAR || interp > FILT_BSPLINE_SIMPLE) return 2;
    const int cubic = interp != LINEAR;
    const int simple = (interp == BSPLINE_SIMPLE || interp == FILT_BSPLINE_SIMPLE);
    const int faithful = (flags & VT_ORACLE_FAITHFUL) != 0;
    const int keep = (flags & VT_ORACLE_KEEP_OUTSIDE) != 0;
    tex_t t = { src, sD, sH, sW, (flags & VT_ORACLE_TEXFRAC8) != 0 };
    const double dims[3] = { (double)gD, (double)sH, (double)sW };

#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t d = 0; d < oD; ++d) {
        for (int64_t h = 0; h < oH; ++h) {
            for (int64_t w = 0; w < oW; ++w) {
                float* dst = out + (d * oH + h) * oW + w;
                const double gd = (double)(d + out_plane0);
                float val;
                if (faithful) {
                    /* transforms.py:265-274: float4 voxf; ndx = dot(voxf, xform[r]) + .5f, all float32.
                     * helper_math.h:1256 dot(float4,float4) = a.x*b.x + a.y*b.y + a.z*b.z + a.w*b.w */
                    float c[3];
                    int outside = 0;
                    for (int r = 0; r < 3; ++r) {
                        const float m0 = (float)m[4 * r], m1 = (float)m[4 * r + 1], m2 = (float)m[4 * r + 2], m3 = (float)m[4 * r + 3];
                        float dot = (float)gd * m0 + (float)h * m1 + (float)w * m2 + 1.0f * m3;
                        c[r] = dot + 0.5f;
                        if (c[r] < 0 || c[r] >= (float)dims[r]) outside = 1;
                    }
                    if (outside) { if (!keep) *dst = 0.0f; continue; }
                    const float cz = c[0] - (float)plane0, cy = c[1], cx = c[2];
                    if (!cubic) val = tex3d_linear(&t, cx, cy, cz);
                    else if (!simple) val = cubic_tex3d_asis(&t, cx, cy, cz);
                    else {
                        const float gx = cx - 0.5f, gy = cy - 0.5f, gz = cz - 0.5f;
                        const float ix = floorf(gx), iy = floorf(gy), iz = floorf(gz);
                        val = cubic_taps(&t, (int64_t)iz, (int64_t)iy, (int64_t)ix, gz - iz, gy - iy, gx - ix, 1);
                    }
                } else {
                    double s[3];
                    int outside = 0;
                    for (int r = 0; r < 3; ++r) {
                        s[r] = fma(m[4 * r], gd, fma(m[4 * r + 1], (double)h, fma(m[4 * r + 2], (double)w, m[4 * r + 3])));
                        if (!(s[r] + 0.5 >= 0.0) || !(s[r] + 0.5 < dims[r])) outside = 1;
                    }
                    if (outside) { if (!keep) *dst = 0.0f; continue; }
                    s[0] -= (double)plane0;
                    const double fl0 = floor(s[0]), fl1 = floor(s[1]), fl2 = floor(s[2]);
                    const float fz = (float)(s[0] - fl0), fy = (float)(s[1] - fl1), fx = (float)(s[2] - fl2);
                    const int64_t iz = (int64_t)fl0, iy = (int64_t)fl1, ix = (int64_t)fl2;
                    if (!cubic) val = linear_taps(&t, iz, iy, ix, fz, fy, fx);
                    else val = cubic_taps(&t, iz, iy, ix, fz, fy, fx, simple);
                }
                *dst = val;
            }
        }
    }
    return 0;
}

/* Plain form: output shape = input shape, whole volume (the reference's only GPU configuration). */
int vt_oracle_affine(const float* src, int64_t D, int64_t H, int64_t W, const float* m4x4,
                     float* out, int interp, int flags)
{
    double m[16];
    for (int i = 0; i < 16; ++i) m[i] = (double)m4x4[i];
    return vt_oracle_affine_ex(src, D, H, W, 0, D, out, D, H, W, 0, m, interp, flags);
}

/* ---- prefilter: bspline.h:2-54 verbatim semantics, float32 ---- */

static float pole_f(void) { return sqrtf(3.0f) - 2.0f; }   /* helper_math.h:1468 */

static float initial_causal(const float* c, uint32_t n, ptrdiff_t step)
{
    const float Pole = pole_f();
    const uint32_t horizon = n < 12u ? n : 12u;       /* bspline.h:7 */
    float zn = Pole;
    float sum = *c;
    for (uint32_t k = 0; k < horizon; ++k) {
        sum += zn * *c;
        zn *= Pole;
        c += step;
    }
    return sum;
}

/* One line: n samples, `step` floats apart, in place (bspline.h:30-54). */
void vt_oracle_prefilter_line(float* coeffs, uint32_t n, ptrdiff_t step)
{
    const float Pole = pole_f();
    const float Lambda = (1.0f - Pole) * (1.0f - 1.0f / Pole);
    float* c = coeffs;
    float prev;
    *c = prev = Lambda * initial_causal(c, n, step);
    for (uint32_t k = 1; k < n; ++k) {
        c += step;
        *c = prev = Lambda * *c + Pole * prev;
    }
    *c = prev = (Pole / (Pole - 1.0f)) * *c;           /* bspline.h:27 */
    for (int64_t k = (int64_t)n - 2; 0 <= k; --k) {
        c -= step;
        *c = prev = Pole * (prev - *c);
    }
}

/* Three passes in the reference's order X (axis 2), Y (axis 1), Z (axis 0), in place
 * (transforms.py:305-307; kernels bspline.h:58-99). */
void vt_oracle_prefilter(float* vol, int64_t D, int64_t H, int64_t W)
{
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t z = 0; z < D; ++z)
        for (int64_t y = 0; y < H; ++y)
            vt_oracle_prefilter_line(vol + (z * H + y) * W, (uint32_t)W, 1);
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t z = 0; z < D; ++z)
        for (int64_t x = 0; x < W; ++x)
            vt_oracle_prefilter_line(vol + z * H * W + x, (uint32_t)H, W);
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x)
            vt_oracle_prefilter_line(vol + y * W + x, (uint32_t)D, H * W);
}

/* transform() on the reference's GPU path end to end (transforms.py:164-226): prefilter a private copy
 * when interp is filt_*, then the kernel.  scratch must hold D*H*W floats when interp >= FILT_BSPLINE. */
int vt_oracle_transform(const float* src, int64_t D, int64_t H, int64_t W, const float* m4x4,
                        float* out, int interp, int flags, float* scratch)
{
    const float* s = src;
    if (interp == FILT_BSPLINE || interp == FILT_BSPLINE_SIMPLE) {
        if (!scratch) return 3;
        memcpy(scratch, src, (size_t)(D * H * W) * sizeof(float));
        vt_oracle_prefilter(scratch, D, H, W);
        s = scratch;
    }
    return vt_oracle_affine(s, D, H, W, m4x4, out, interp, flags);
}

int vt_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void vt_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
