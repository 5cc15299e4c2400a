// vt_kernels_layout.hip -- axis 0 <-> 2 exchange of a volume (gfx950).
//
// Rotations about array axis 2 ([a b 0; c d 0; 0 0 1]) are axis-0-separable once axes 0 and 2 are exchanged.  The
// marching kernels then run on an exchanged resident copy (built once, lazily) and write an exchanged result, which one
// more pass of this kernel turns back: dst[k][j][i] = src[i][j][k].  LDS tile transpose per j: reads are coalesced along k,
// writes along i; 8 B per voxel of HBM traffic.
#include "vt_internal.h"

namespace vt {

// src element (i, j, k) at i*ss0 + j*ss1 + k ; dst element (k, j, i) at k*ds0 + j*ds1 + i
// 64 x 64 tile, 256 threads.  VEC = 4: both sides move 16 bytes per lane (rows of 256 B); needs n0, n2 and all four strides
// to be multiples of 4 and 16-byte aligned bases.  VEC = 1: any geometry.
template <int VEC>
__global__ __launch_bounds__(256) void transpose02_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int n0, int n1, int n2, int64_t ss0, int64_t ss1, int64_t ds0, int64_t ds1,
                                                           int tiles_k)
{
    __shared__ float tile[64][65];
    const int tk = blockIdx.x % tiles_k, ti = blockIdx.x / tiles_k;
    const int j = blockIdx.y;
    const int k0 = tk * 64, i0 = ti * 64;
    if constexpr (VEC == 4) {
        const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;  // 16 x 16 threads, 4 floats each
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + ly + 16 * r, k = k0 + 4 * lx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n0 && k < n2) v = *reinterpret_cast<const float4*>(src + (int64_t)i * ss0 + (int64_t)j * ss1 + k);
            tile[ly + 16 * r][4 * lx + 0] = v.x; tile[ly + 16 * r][4 * lx + 1] = v.y;
            tile[ly + 16 * r][4 * lx + 2] = v.z; tile[ly + 16 * r][4 * lx + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + ly + 16 * r, i = i0 + 4 * lx;
            if (k < n2 && i < n0) {
                const float4 v = make_float4(tile[4 * lx + 0][ly + 16 * r], tile[4 * lx + 1][ly + 16 * r],
                                             tile[4 * lx + 2][ly + 16 * r], tile[4 * lx + 3][ly + 16 * r]);
                *reinterpret_cast<float4*>(dst + (int64_t)k * ds0 + (int64_t)j * ds1 + i) = v;
            }
        }
    } else {
        const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;  // 64 x 4 threads
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + ly + 4 * r, k = k0 + lx;
            tile[ly + 4 * r][lx] = (i < n0 && k < n2) ? src[(int64_t)i * ss0 + (int64_t)j * ss1 + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + ly + 4 * r, i = i0 + lx;
            if (k < n2 && i < n0) dst[(int64_t)k * ds0 + (int64_t)j * ds1 + i] = tile[lx][ly + 4 * r];
        }
    }
}

hipError_t launch_transpose02(const float* src, float* dst, int n0, int n1, int n2, int64_t ss0, int64_t ss1,
                              int64_t ds0, int64_t ds1, hipStream_t stream)
{
    const int tiles_k = (n2 + 63) / 64, tiles_i = (n0 + 63) / 64;
    const int64_t gx = (int64_t)tiles_k * tiles_i;
    if (gx > 0x7fffffffLL || n1 > 65535 || n1 <= 0) return hipErrorInvalidValue;
    const bool vec = (n0 % 4 == 0) && (n2 % 4 == 0) && (ss0 % 4 == 0) && (ss1 % 4 == 0) && (ds0 % 4 == 0) && (ds1 % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(transpose02_kernel<4>, dim3((unsigned)gx, (unsigned)n1), dim3(256), 0, stream, src, dst, n0, n1, n2, ss0, ss1,
                           ds0, ds1, tiles_k);
    else
        hipLaunchKernelGGL(transpose02_kernel<1>, dim3((unsigned)gx, (unsigned)n1), dim3(256), 0, stream, src, dst, n0, n1, n2, ss0, ss1,
                           ds0, ds1, tiles_k);
    return hipGetLastError();
}

// dense [D][H][W] -> padded resident layout [(D+2R)][(H+2R)][P]: the volume with R voxels of whole-sample-symmetric ("mirror")
// extension on every side (index -1 <- 1, n <- n-2, reflected again for extents below R), columns beyond W+2R zero.
// scipy's spline interpolation extends its coefficients this way for taps of in-range coordinates (VT_EDGE_SCIPY).
__device__ __forceinline__ int reflect_index(int i, int n)
{
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

__global__ __launch_bounds__(256) void mirror_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int H, int W, int R, int P)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, z = blockIdx.z;
    if (x >= P) return;
    float v = 0.0f;
    if (x < W + 2 * R)
        v = src[((int64_t)reflect_index(z - R, D) * H + reflect_index(y - R, H)) * W + reflect_index(x - R, W)];
    dst[((int64_t)z * (H + 2 * R) + y) * P + x] = v;
}

hipError_t launch_mirror_pad(const float* src, float* dst, int D, int H, int W, int R, int P, hipStream_t stream)
{
    const dim3 grid((P + 255) / 256, H + 2 * R, D + 2 * R);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mirror_pad_kernel, grid, dim3(256), 0, stream, src, dst, D, H, W, R, P);
    return hipGetLastError();
}

}  // namespace vt
