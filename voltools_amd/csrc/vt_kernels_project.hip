// vt_kernels_project.hip -- axis-0 projection of a transformed volume (rotate, then sum(axis=0);
// examples/projections.py:20-26 of the reference does the sum with cupy after the transform).
//
// For matrices of the axis-0-separable form [1 0 0 tz; 0 a b ty; 0 c d tx] the transform factorises:
//     out[d, h, w] = sum_k wz[k] * P_{d + zoff - halo + k}(h, w),       P_z = in-plane interpolation of source plane z
// so   sum_d out[d, h, w] = I( sum_z c_z * src[z] )(h, w)   with   c_z = sum over the valid output planes d that tap z
// (interpolation is linear in the data).  One streaming pass over the resident source (4 B/voxel, no store of the
// transformed volume) builds S = sum_z c_z src[z]; the 2-D interpolation of S is a launch of the ordinary transform
// kernels on a 3-plane helper volume [S, S, S] (the z weights of an integer position sum to 1 over planes 0..2).
// General matrices: transform into scratch, then the same pass with c_z = 1 over the transformed volume.
#include "vt_internal.h"

namespace vt {

// dst[y, x] (+ optional copies at dst + k * dst_plane) = sum_z c_z * src[z, y, x]
// Workgroup = 64 column groups x 4 z-slices (z = slice, slice + 4, ...); the four partial sums meet in LDS and are added
// in a fixed order (deterministic).  VEC = 4: rows are 16-byte aligned (pitch % 4 == 0), one float4 per lane.
template <int VEC>
__global__ __launch_bounds__(256) void plane_sum_kernel(const float* __restrict__ src, float* __restrict__ dst, const ProjectParams q)
{
    __shared__ float part[3][64][VEC];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * 64 + lane;          // column group index over (y, xv)
    const int y = (int)(g / q.nxv), xv = (int)(g - (int64_t)y * q.nxv);
    const bool active = y < q.H;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    if (active) {
        const float* p = src + ((int64_t)slice * q.H + y) * q.src_pitch + (int64_t)xv * VEC;
        const int64_t step = (int64_t)4 * q.H * q.src_pitch;
#pragma unroll 4
        for (int z = slice; z < q.D; z += 4, p += step) {
            float c = 1.0f;
            if (!q.uniform) {
                // plane z is tapped with weight wz[k] by output plane d = z - zoff + halo - k; count the valid ones
                c = 0.f;
                const int d0 = z - q.zoff + q.halo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int d = d0 - k;
                    c += (k < q.ntap && d >= q.dlo && d <= q.dhi) ? q.wz[k] : 0.f;
                }
            }
            if constexpr (VEC == 4) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v vv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));         // read once: streaming load
                const float4 v = make_float4(vv.x, vv.y, vv.z, vv.w);
                acc[0] = fmaf(c, v.x, acc[0]); acc[1] = fmaf(c, v.y, acc[1]);
                acc[2] = fmaf(c, v.z, acc[2]); acc[3] = fmaf(c, v.w, acc[3]);
            } else {
                acc[0] = fmaf(c, p[0], acc[0]);
            }
        }
    }
    if (slice > 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) part[slice - 1][lane][i] = acc[i];
    }
    __syncthreads();
    if (slice == 0 && active) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = ((acc[i] + part[0][lane][i]) + part[1][lane][i]) + part[2][lane][i];
        float* o = dst + (int64_t)y * q.dst_pitch + (int64_t)xv * VEC;
        for (int k = 0; k < q.copies; ++k, o += q.dst_plane) {
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            else o[0] = acc[0];
        }
    }
}

hipError_t launch_plane_sum(const float* src, float* dst, const ProjectParams& q, hipStream_t stream)
{
    const int64_t groups = (int64_t)q.H * q.nxv;
    const int64_t blocks = (groups + 63) / 64;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (q.vec == 4) hipLaunchKernelGGL(plane_sum_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, q);
    else hipLaunchKernelGGL(plane_sum_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, q);
    return hipGetLastError();
}

}  // namespace vt
