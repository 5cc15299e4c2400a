// vt_kernels_quad.hip -- marching transform kernel on the plane-QUAD layout (gfx950), both interpolation families.
//
// Same problem as vt_kernels_march.hip (matrices [1 0 0 tz; 0 a b ty; 0 c d tx]: the reference's `transform` kernel body,
// transforms.py:253-281, with linearTex3D / cubicTex3D / cubicTex3DSimple of helper_interpolation.h:3-68 as the sampler),
// re-cut around what the round-1 counters showed: that kernel issued one scalar instruction per vector instruction
// (ring-slot multiplies, per-plane validity branches, exec save/restore around every staging load, 64-bit address
// arithmetic per store) and 95 instructions per wave and plane for 128 voxels.  Here:
//
//   * Resident layout [z/4][y][x][4] (relayout_zquad, built once per handle): the four source planes of a quad are
//     interleaved per position, so ONE ds_read_b128 returns a tap for four planes (256 B/clk/CU, half the LDS instructions of
//     the pair layout, a quarter of the plain one), the in-plane sums of four planes are packed-f32 FMAs, and a staged row
//     of the footprint is a contiguous run of 16-byte vectors in HBM (a 16 x 32 tile at 0 degrees: 544-byte runs instead
//     of 144-byte ones).  Every position is its own vector: row spans are packed with no alignment waste.
//   * One step = one quad = four output planes: half the barriers per voxel of the pair kernel.  Two ring slots, the next
//     quad's loads are issued right after the barrier; the counted wait before the barrier is an immediate
//     (vmcnt(4 * NPIX): only the previous step's stores are younger than the loads being waited for).
//   * Output through a buffer descriptor based at the tile's first voxel: the per-pixel offset is a loop-invariant VGPR,
//     the plane offset one scalar add.  The ring slot toggles with one scalar XOR.
//   * Steps whose four outputs all lie in the chunk, on tiles wholly inside the source and the output, run a
//     predicate-free body; everything else (first / last step of a chunk, tiles cut by the skirt or the output edge)
//     takes the general body.
// Per output voxel (trilinear): one ds_read_b128, ~5 VALU, 1 store, < 0.1 scalar instructions.
//
// Round 3:
//   * KIND 3 = trilinear with an INTEGER axis-0 offset (fz == 0: every rotation about axis 0, every in-plane map).  The weight of
//     the second tap plane is exactly 0, so output plane d is the in-plane interpolation of source plane d + zoff alone
//     (fmaf(0, b - a, a) == a for finite data: bit-identical to KIND 0).  No carried partial, and a chunk of dch output planes reads
//     exactly dch / 4 quads instead of dch / 4 + 1: the history quad was 1/8 of a 32-plane chunk's staged bytes.
//   * Lanes are assigned to pixels so that every 16-lane service group of ds_read_b128 (MI355X_MICROARCH.md, LDS:
//     {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) holds 16 CONSECUTIVE pixels of a pixel row: inside one source row their taps are
//     then consecutive 16-byte slots whatever the angle (the identity mapping spreads a group over 28 pixels, i.e. 24 slots at
//     30 degrees: conflicts by construction).  [model, tools/quad_conflict_sim.py] with bank-aware row starts: 1.67 -> 1.12
//     LDS cycles per read at 15 degrees, 2.0 -> 1.31 at 20.  The store of a wave still covers the same two 128-byte row segments.
#include "vt_internal.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include "vt_device.h"
#include "vt_march_common.h"

#include <type_traits>

namespace vt {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kQuadMaxIt = 5;     // staging instructions per thread and quad: footprints (with row padding) up to 5 * NT vectors

// plain [z][y][P] -> quad layout [z/4][y][Pq]: element (z, y, x) at 4x + (z & 3) of row ((z >> 2), y); positions W .. Wq-1
// of every row stay zero (the border colour; the staging loads fetch position W of row 0 for every out-of-volume vector)
__global__ __launch_bounds__(256) void relayout_zquad(const float* __restrict__ src, float* __restrict__ dst,
                                                       int D, int H, int W, int P, int Pq)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int qq = blockIdx.z;
    if (4 * x >= Pq) return;
    if (x >= W) {                                  // the row's pad positions: written here, the buffer is not cleared beforehand (round 5)
        *reinterpret_cast<v4f*>(dst + ((int64_t)qq * H + y) * Pq + 4 * (int64_t)x) = v4f{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int z0 = 4 * qq;
    const int64_t plane = (int64_t)H * P;
    const float* s = src + ((int64_t)z0 * H + y) * P + x;
    v4f v;
    v.x = s[0];
    v.y = (z0 + 1 < D) ? s[plane] : 0.0f;
    v.z = (z0 + 2 < D) ? s[2 * plane] : 0.0f;
    v.w = (z0 + 3 < D) ? s[3 * plane] : 0.0f;
    *reinterpret_cast<v4f*>(dst + ((int64_t)qq * H + y) * Pq + 4 * (int64_t)x) = v;
}

hipError_t launch_relayout_zquad(const float* src, float* dst, int D, int H, int W, int P, int Pq, hipStream_t stream)
{
    const dim3 grid((Pq / 4 + 255) / 256, H, (D + 3) / 4);
    if (grid.y > 65535 || grid.z > 65535 || (Pq & 3) || Pq < 4 * W) return hipErrorInvalidValue;
    hipLaunchKernelGGL(relayout_zquad, grid, dim3(256), 0, stream, src, dst, D, H, W, P, Pq);
    return hipGetLastError();
}

// Round 4: the plane-quad copy of the Z-CONVOLVED volume, e[z] = w0 c[z-1] + w1 c[z] + w2 c[z+1] with the cubic B-spline weights of
// fraction 0 (1/6, 2/3, 1/6; the fourth is exactly 0) and zero border.  A cubic launch whose axis-0 offset is an integer -- every rotation
// about axis 0, every in-plane map: the README sweep -- weighs its four tap planes with exactly these numbers, whatever the matrix, so
// the axis-0 part of its 64-tap sum can be formed ONCE per volume: output plane d is then the in-plane 16-tap interpolation of plane
// d + zoff of this copy (KIND 4 below: no carried partials, no history quad, 16-plane chunks -- the trilinear KIND 3's structure with
// cubic in-plane taps).  Same 64 products as cubicTex3D (helper_interpolation.h:8-40), summed z first instead of x first.
__global__ __launch_bounds__(256) void relayout_zquad_fir(const float* __restrict__ src, float* __restrict__ dst,
                                                           int D, int H, int W, int P, int Pq, int simple)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int qq = blockIdx.z;
    if (4 * x >= Pq) return;
    if (x >= W) {
        *reinterpret_cast<v4f*>(dst + ((int64_t)qq * H + y) * Pq + 4 * (int64_t)x) = v4f{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int z0 = 4 * qq;
    const int64_t plane = (int64_t)H * P;
    const float* s = src + ((int64_t)z0 * H + y) * P + x;
    float wz[4];
    if (simple) cubic_weights<true>(0.0f, wz); else cubic_weights<false>(0.0f, wz);
    float c[6];                                   // planes z0 - 1 .. z0 + 4
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int z = z0 - 1 + k;
        c[k] = (z >= 0 && z < D) ? s[(int64_t)(k - 1) * plane] : 0.0f;
    }
    v4f v;
    // the association of the marching kernel's axis-0 combination (zcombine): w0 first, then two FMAs
    v.x = fmaf(wz[2], c[2], fmaf(wz[1], c[1], wz[0] * c[0]));
    v.y = (z0 + 1 < D) ? fmaf(wz[2], c[3], fmaf(wz[1], c[2], wz[0] * c[1])) : 0.0f;
    v.z = (z0 + 2 < D) ? fmaf(wz[2], c[4], fmaf(wz[1], c[3], wz[0] * c[2])) : 0.0f;
    v.w = (z0 + 3 < D) ? fmaf(wz[2], c[5], fmaf(wz[1], c[4], wz[0] * c[3])) : 0.0f;
    *reinterpret_cast<v4f*>(dst + ((int64_t)qq * H + y) * Pq + 4 * (int64_t)x) = v;
}

hipError_t launch_relayout_zquad_fir(const float* src, float* dst, int D, int H, int W, int P, int Pq, bool simple, hipStream_t stream)
{
    const dim3 grid((Pq / 4 + 255) / 256, H, (D + 3) / 4);
    if (grid.y > 65535 || grid.z > 65535 || (Pq & 3) || Pq < 4 * W) return hipErrorInvalidValue;
    hipLaunchKernelGGL(relayout_zquad_fir, grid, dim3(256), 0, stream, src, dst, D, H, W, P, Pq, simple ? 1 : 0);
    return hipGetLastError();
}

// Round 5: the plane-quad copy (FIR: of the z-convolved volume) of the IN-PLANE TRANSPOSED orientation straight from the handle's plain
// copy -- dst row (quad, x) holds source column x: element (z, y, x) at 4y + (z & 3) -- instead of transpose02 into an exchanged plain copy
// and relayout_zquad[_fir] from there.  Same values bit for bit (the FIR sums the same three planes in the same association), one read and
// one write of the volume instead of two, and no volume-sized temporary: under a resident budget (vt_volume_set_max_resident) a README sweep
// rebuilds this copy once per half turn, and the temporary's hipMalloc / hipFree cost as much as its kernel.
// A workgroup transposes a 32 (y) x 64 (x) tile of one quad through LDS: reads coalesced along x, writes 512 contiguous bytes per half wave
// along y.  Rows are padded by one vector: lane y reads bank group 4y mod 64, no conflict among the 16 lanes of a b128 pass.
constexpr int kSwapTY = 32, kSwapTX = 64;
template <bool FIR>
__global__ __launch_bounds__(256) void relayout_zquad_swap12(const float* __restrict__ src, float* __restrict__ dst,
                                                              int D, int H, int W, int P, int Pq, int simple)
{
    __shared__ v4f tile[kSwapTY * (kSwapTX + 1)];
    const int x0 = blockIdx.x * kSwapTX, y0 = blockIdx.y * kSwapTY, qq = blockIdx.z;
    const int z0 = 4 * qq;
    const int64_t plane = (int64_t)H * P;
    float wz[4];
    if (simple) cubic_weights<true>(0.0f, wz); else cubic_weights<false>(0.0f, wz);
    const int xl = threadIdx.x & (kSwapTX - 1);
#pragma unroll
    for (int i = 0; i < kSwapTY / 4; ++i) {
        const int yl = (threadIdx.x >> 6) + 4 * i;
        const int x = x0 + xl, y = y0 + yl;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (x < W && y < H) {
            const float* s = src + ((int64_t)z0 * H + y) * P + x;
            if (FIR) {
                float c[6];                           // planes z0 - 1 .. z0 + 4
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int z = z0 - 1 + k;
                    c[k] = (z >= 0 && z < D) ? s[(int64_t)(k - 1) * plane] : 0.0f;
                }
                v.x = fmaf(wz[2], c[2], fmaf(wz[1], c[1], wz[0] * c[0]));
                v.y = (z0 + 1 < D) ? fmaf(wz[2], c[3], fmaf(wz[1], c[2], wz[0] * c[1])) : 0.0f;
                v.z = (z0 + 2 < D) ? fmaf(wz[2], c[4], fmaf(wz[1], c[3], wz[0] * c[2])) : 0.0f;
                v.w = (z0 + 3 < D) ? fmaf(wz[2], c[5], fmaf(wz[1], c[4], wz[0] * c[3])) : 0.0f;
            } else {
                v.x = s[0];
                v.y = (z0 + 1 < D) ? s[plane] : 0.0f;
                v.z = (z0 + 2 < D) ? s[2 * plane] : 0.0f;
                v.w = (z0 + 3 < D) ? s[3 * plane] : 0.0f;
            }
        }
        tile[yl * (kSwapTX + 1) + xl] = v;
    }
    __syncthreads();
    const int yl = threadIdx.x & (kSwapTY - 1);
#pragma unroll
    for (int i = 0; i < kSwapTX / 8; ++i) {
        const int xr = (threadIdx.x >> 5) + 8 * i;
        const int x = x0 + xr, y = y0 + yl;
        if (x < W && 4 * y < Pq)                      // y >= H: the row's pad positions (the tile holds zeros there)
            *reinterpret_cast<v4f*>(dst + ((int64_t)qq * W + x) * Pq + 4 * (int64_t)y) = tile[yl * (kSwapTX + 1) + xr];
    }
}

// D, H, W, P: the handle's plain copy; the copy's rows are the W source columns, Pq floats each (>= 4 * H)
hipError_t launch_relayout_zquad_swap12(const float* src, float* dst, int D, int H, int W, int P, int Pq, bool fir, bool simple, hipStream_t stream)
{
    const dim3 grid((W + kSwapTX - 1) / kSwapTX, (Pq / 4 + kSwapTY - 1) / kSwapTY, (D + 3) / 4);
    if (grid.y > 65535 || grid.z > 65535 || (Pq & 3) || Pq < 4 * H) return hipErrorInvalidValue;
    if (fir) hipLaunchKernelGGL(relayout_zquad_swap12<true>, grid, dim3(256), 0, stream, src, dst, D, H, W, P, Pq, simple ? 1 : 0);
    else hipLaunchKernelGGL(relayout_zquad_swap12<false>, grid, dim3(256), 0, stream, src, dst, D, H, W, P, Pq, 0);
    return hipGetLastError();
}

__device__ __forceinline__ int floordiv4(int a) { return a >> 2; }           // arithmetic shift = floor division

// One output column (all planes of a chunk at one in-plane pixel) gathered straight from the quad copy -- element (z, y, x)
// sits at ((z >> 2) * H + y) * Pq + 4x + (z & 3).  The cold path of affine_march4: footprints larger than the planned slot.
template <int KIND>
__device__ __forceinline__ void quad_gather_column(const float* __restrict__ srcq, float* __restrict__ out, const AffineParams& p, int64_t oo,
                                                int d_begin, int d_end, bool in_yx, int gy0, int gx0, float fy, float fx)
{
    constexpr bool CUBIC = KIND == 1 || KIND == 2 || KIND == 4;
    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;
    auto fetch = [&](int z, int y, int x) -> float {
        if ((unsigned)z < (unsigned)p.sD && (unsigned)y < (unsigned)p.sH && (unsigned)x < (unsigned)p.sW)
            return srcq[((int64_t)(z >> 2) * p.sH + y) * p.sPq + 4 * x + (z & 3)];
        return 0.f;
    };
    float wx[4] = {0.f, 0.f, 0.f, 0.f}, wy[4] = {0.f, 0.f, 0.f, 0.f}, wz[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (KIND == 4) {
        if (p.flags & (1 << 18)) { cubic_weights<true>(fx, wx); cubic_weights<true>(fy, wy); } else { cubic_weights<false>(fx, wx); cubic_weights<false>(fy, wy); }
    } else if constexpr (CUBIC) { cubic_weights<KIND == 2>(fx, wx); cubic_weights<KIND == 2>(fy, wy); cubic_weights<KIND == 2>(p.fz, wz); }
    for (int d = d_begin; d < d_end; ++d) {
        const double ez = (double)d + p.m[3];
        const bool inside = in_yx && (ez >= p.vlo[0]) && (ez < p.vhi[0]);
        float val = 0.f;
        if (inside) {
            if constexpr (!CUBIC) {
                float pl[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int z = d + p.zoff + c;
                    const float a00 = fetch(z, gy0, gx0), a01 = fetch(z, gy0, gx0 + 1), a10 = fetch(z, gy0 + 1, gx0), a11 = fetch(z, gy0 + 1, gx0 + 1);
                    const float x0 = fmaf(fx, a01 - a00, a00);
                    const float x1 = fmaf(fx, a11 - a10, a10);
                    pl[c] = fmaf(fy, x1 - x0, x0);
                }
                val = fmaf(p.fz, pl[1] - pl[0], pl[0]);
            } else if constexpr (KIND == 4) {
                // the z-convolved copy: one plane, 16 in-plane taps
                const int z = d + p.zoff;
                float accy = 0.f;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int y = gy0 - 1 + bb;
                    float accx = wx[0] * fetch(z, y, gx0 - 1);
                    accx = fmaf(wx[1], fetch(z, y, gx0), accx);
                    accx = fmaf(wx[2], fetch(z, y, gx0 + 1), accx);
                    accx = fmaf(wx[3], fetch(z, y, gx0 + 2), accx);
                    accy = fmaf(wy[bb], accx, accy);
                }
                val = accy;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int z = d + p.zoff - 1 + c;
                    float accy = 0.f;
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const int y = gy0 - 1 + bb;
                        float accx = wx[0] * fetch(z, y, gx0 - 1);
                        accx = fmaf(wx[1], fetch(z, y, gx0), accx);
                        accx = fmaf(wx[2], fetch(z, y, gx0 + 1), accx);
                        accx = fmaf(wx[3], fetch(z, y, gx0 + 2), accx);
                        accy = fmaf(wy[bb], accx, accy);
                    }
                    val = (c == 0) ? wz[0] * accy : fmaf(wz[c], accy, val);
                }
            }
            out[oo + (int64_t)d * p.ostride] = val;
        } else if (!keep) out[oo + (int64_t)d * p.ostride] = 0.0f;
    }
}

// pixel position (0..63 within the wave's 64 pixels) of a lane, and its inverse: each ds_read_b128 service group gets 16 consecutive
// positions.  Piecewise shifts of 4-lane blocks inside each half-wave: lanes 0-3 | 4-11 | 12-15 | 16-19 | 20-27 | 28-31 go to
// positions 0-3 | 16-23 | 4-7 | 24-27 | 8-15 | 28-31.
__device__ __forceinline__ int quad_lane_to_pos(int lane)
{
    const int blk = (lane >> 2) & 7;                       // 4-lane block inside the half-wave
    // shift in units of 4 positions, biased by +3, one nibble per block: {0, +3, +3, -2, +2, -3, -3, 0}
    const int sh = (int)((0x30051663u >> (4 * blk)) & 15u) - 3;
    return lane + 4 * sh;
}
__device__ __forceinline__ int quad_pos_to_lane(int pos)
{
    const int blk = (pos >> 2) & 7;
    // inverse shifts: {0, +2, +3, +3, -3, -3, -2, 0}
    const int sh = (int)((0x31006653u >> (4 * blk)) & 15u) - 3;
    return pos + 4 * sh;
}

template <int KIND, int TH, int TW, int NT>
// (KIND 3 with 512 threads is held to three workgroups per CU by its 100+ scalar registers -- 7 waves per SIMD; capped at 96 it runs four
// and is 0.7 % slower: 512^3 0.1748 against 0.1731 ms, 1024^3 1.497 against 1.486, one process per variant)
__global__ __launch_bounds__(NT, (KIND == 4 && TH * TW == 2 * NT) ? 4 : 1) void affine_march4(const float* __restrict__ srcq, float* __restrict__ out, const AffineParams p)
{
    static_assert(NT % TW == 0 && TH % (NT / TW) == 0 && NT % 64 == 0 && 64 % TW == 0, "tile/thread mapping");
    constexpr bool CUBIC = KIND == 1 || KIND == 2 || KIND == 4;      // in-plane stencil: 4 x 4 (else 2 x 2)
    constexpr bool ZID = KIND == 3 || KIND == 4;  // integer axis-0 offset: ONE tap plane per output plane (KIND 3: trilinear, the other plane has
                                                  // weight 0; KIND 4: cubic on the z-convolved copy, relayout_zquad_fir)
    constexpr int HALO = CUBIC ? 1 : 0;           // in-plane halo
    constexpr int ZH = ZID ? 0 : HALO;            // axis-0 halo
    constexpr int NR = 2 + 2 * HALO;              // tap rows (and columns) per pixel
    constexpr int NC = 2 * ZH + 1;                // carried in-plane partials per pixel (unused for ZID)
    constexpr int ZNEW = ZID ? 0 : 1;             // the newest tap plane of output d is d + zoff + ZH + ZNEW
    constexpr int RP = NT / TW;
    constexpr int NPIX = TH / RP;
    constexpr int NSTORE = 4 * NPIX;              // store instructions of a full step (one per pixel and plane)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
#ifdef VT_EXPERIMENTS      // VT_EXP_NOLOOP + VT_EXP_NOLDS: the launch alone (workgroup dispatch, LDS allocation, argument loads)
    if ((p.flags & (1 << 27)) && (p.flags & (1 << 26))) return;
    long long stamp[5];                           // VT_EXP_NOLOOP: cycle stamps of the set-up's phases (tools/setup_phases.py)
    stamp[0] = clock64();
#endif
    int tw_i, th_i, chunk;
    if (p.flags & (1 << 29)) {
        // 2-D grid: blockIdx.y = chunk, blockIdx.x = in-plane tile.  Workgroups are dispatched x-fastest, so each XCD gets a
        // contiguous band of every chunk layer in turn: the chip works on one or two layers at a time (compact write stream,
        // halos shared inside the band), and the tile decode needs one multiply-high instead of two integer divisions.
        chunk = (p.flags & (1 << 20)) ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;      // odd launches: layers from the last to the first
        unsigned u = (unsigned)xcd_contiguous(blockIdx.x, gridDim.x);
        if (p.flags & (1 << 30)) u = gridDim.x - 1 - u;             // A/B: tiles walked in descending order
        if (p.flags & (1 << 24)) {                // h fastest (in-plane transposed copy)
            tw_i = (int)__umulhi(u, p.nTh_magic);
            th_i = (int)u - tw_i * p.nTh;
        } else {
            th_i = (int)__umulhi(u, p.nTw_magic);
            tw_i = (int)u - th_i * p.nTw;
        }
    } else {
        int t = xcd_contiguous(blockIdx.x, gridDim.x);
        if (p.flags & (1 << 20)) t = (int)gridDim.x - 1 - t;
        march_tile(p, t, th_i, tw_i, chunk);
    }
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int d_begin = chunk ? chunk * p.dch + p.dshift : 0;
    const int d_end = min((chunk + 1) * p.dch + p.dshift, p.oD);

    // in-plane footprint (rows 1, 2; column 0 of the matrix is zero)
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
    const double z_lo = (double)d_begin + p.m[3], z_hi = (double)(d_end - 1) + p.m[3];
    any_valid = any_valid && (z_hi >= p.vlo[0] - kTileMargin) && (z_lo < p.vhi[0] + kTileMargin);
    all_valid = all_valid && (z_lo >= p.vlo[0] + kTileMargin) && (z_hi < p.vhi[0] - kTileMargin);
    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;
    const int64_t ostride = p.ostride, orow = p.orow;     // element strides of an output plane / row (axis swaps)
    // lane -> pixel: position `pos` of the wave's 64 pixels (TW | 64: a pixel row never straddles a wave)
    const bool lane_perm = (p.flags & (1 << 23)) != 0;
    const int lane_id = tid & 63;
    const int pos = lane_perm ? quad_lane_to_pos(lane_id) : lane_id;
    const int kw = pos % TW;
    const int jh0 = (tid >> 6) * (64 / TW) + pos / TW;
    // lanes holding the neighbouring pixels of the pixel row (used where kw > 0 / kw < TW - 1 only)
    const int lane_left = lane_perm ? quad_pos_to_lane((pos + 63) & 63) : ((lane_id + 63) & 63);
    const int lane_right = lane_perm ? quad_pos_to_lane((pos + 1) & 63) : ((lane_id + 1) & 63);

    if (!any_valid) {
        if (!keep) {
#pragma unroll
            for (int px = 0; px < NPIX; ++px) {
                const int h = h0 + jh0 + px * RP, w = w0 + kw;
                if (h < p.oH && w < p.oW) {
                    float* optr = out + (int64_t)d_begin * ostride + (int64_t)h * orow + w;
                    for (int d = d_begin; d < d_end; ++d, optr += ostride) *optr = 0.0f;
                }
            }
        }
        return;
    }

    // tile-uniform, but float64 -> int conversion is a vector instruction: without the explicit move both floors sit in vector registers
    // for the whole kernel (the compiler drops __builtin_amdgcn_readfirstlane of a value it knows to be uniform)
    int o1, o2;
    {
        const int f1 = (int)floor(lo[1]), f2 = (int)floor(lo[2]);
        // The builtin on a value the compiler can no longer prove uniform (it passed through an empty asm statement): the compiler emits
        // v_readfirstlane_b32 itself, with the wait states its hazard recogniser knows.  (Round 4 wrote the instruction as inline asm with
        // hand-counted s_nop on both sides -- the recogniser does not look into asm statements, and the row kernel, vt_kernels_rows.hip, got
        // stale origins from that pattern without the s_nop.  The builtin alone, on a value known to be uniform, is dropped, and both floors
        // then sit in vector registers for the whole kernel.)
        int g1 = f1, g2 = f2;
        asm volatile("" : "+v"(g1), "+v"(g2));
        o1 = __builtin_amdgcn_readfirstlane(g1);
        o2 = __builtin_amdgcn_readfirstlane(g2);
        o1 -= HALO;
        o2 -= HALO;                               // every position is its own 16-byte vector: no alignment of the origin
    }

    // ---- per-pixel tap geometry ----
    int iy[NPIX], ix[NPIX];
    float fy[NPIX], fx[NPIX];
    float wy[NPIX][4], wx[NPIX][4];
    bool in_yx[NPIX], pix_ok[NPIX];
    int ob[NPIX];                                 // byte offset of the pixel relative to the tile's first voxel
    const double by = base[1] - (double)o1, bx = base[2] - (double)o2;
#pragma unroll
    for (int px = 0; px < NPIX; ++px) {
        const int j = jh0 + px * RP;
        const double sy = fma(p.m[5], (double)j, fma(p.m[6], (double)kw, by));
        const double sx = fma(p.m[9], (double)j, fma(p.m[10], (double)kw, bx));
        const double fyd = floor(sy), fxd = floor(sx);
        fy[px] = (float)(sy - fyd);
        fx[px] = (float)(sx - fxd);
        iy[px] = (int)fyd;
        ix[px] = (int)fxd;
        if constexpr (KIND == 4) {
            if (p.flags & (1 << 18)) { cubic_weights<true>(fy[px], wy[px]); cubic_weights<true>(fx[px], wx[px]); }
            else { cubic_weights<false>(fy[px], wy[px]); cubic_weights<false>(fx[px], wx[px]); }
        } else if constexpr (CUBIC) { cubic_weights<KIND == 2>(fy[px], wy[px]); cubic_weights<KIND == 2>(fx[px], wx[px]); }
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { wy[px][k] = 0.f; wx[px][k] = 0.f; }
        }
        // rows 1, 2 of an axis-0-separable matrix ignore d
        in_yx[px] = all_valid || (canonical_inside_axis(p, 1, 0, h0 + j, w0 + kw) && canonical_inside_axis(p, 2, 0, h0 + j, w0 + kw));
        pix_ok[px] = (h0 + j < p.oH) && (w0 + kw < p.oW);
        ob[px] = (int)(((int64_t)j * orow + kw) * 4);           // < 2^31 (host-checked)
    }

#ifdef VT_EXPERIMENTS
    stamp[1] = clock64();
#endif
    // ---- row spans of the footprint, packed (one vector per position) ----
    // Exact and integer-only: every pixel taps columns [ix - HALO, ix + HALO + 1] of rows iy - HALO .. iy + HALO + 1 of the
    // box.  Along a pixel row (lanes kw = 0 .. TW-1 of equal j) iy and ix are monotone in kw, so the lanes that tap a given
    // source row form a run whose extreme columns sit at the run's two ends: only lanes at a run boundary record their
    // columns (LDS min / max), which keeps same-address contention at a handful of lanes at every angle.  One wave then
    // turns (min, max) per row into (span start, first vector) by a 64-entry prefix sum and writes each vector's row index.
    // (The first version computed the spans analytically in float64 on one wave, march_row_span -- 200 float64 operations on the
    // critical path of every workgroup while three waves waited; the host still sizes the slot with that superset.)
    int voff[kQuadMaxIt];                         // byte offset inside a quad-plane of each vector this thread stages
    int q[NPIX][NR];                              // byte offset of the first tap of each tap row inside a ring slot
    int nvec;
    {
        int* tab = reinterpret_cast<int*>(lds);   // [0..63] column min -> span start, [64..127] column max -> first vector, [128] total
        unsigned char* vrow = reinterpret_cast<unsigned char*>(tab + kTabInts);
#ifdef VT_QUAD_ANALYTIC_SPANS                     // A/B: the float64 polygon clipping of the plain / pair marching kernels
        build_span_table<TH, TW, HALO, 1>(tab, p, p.Ly, by, bx, tid);
#else
        if (tid < 2 * kRowsMax) tab[tid] = (tid < kRowsMax) ? 0x7fffffff : (int)0x80000000;
        __syncthreads();
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
            // neighbours along the pixel row (TW divides 64: a pixel row never straddles a wave)
            const int iy_l = __shfl(iy[px], lane_left), iy_r = __shfl(iy[px], lane_right);
            const bool edge = (kw == 0) || (kw == TW - 1) || (iy_l != iy[px]) || (iy_r != iy[px]);
            if (edge) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int row = min(max(iy[px] - HALO + r, 0), kRowsMax - 1);
                    atomicMin(&tab[row], ix[px] - HALO);
                    atomicMax(&tab[kRowsMax + row], ix[px] + HALO + 1);
                }
            }
        }
        __syncthreads();
#ifdef VT_EXPERIMENTS
        stamp[2] = clock64();
#endif
        if (tid < kRowsMax) {
            const int lane = tid;
            const int mn = tab[lane], mx = tab[kRowsMax + lane];
            const bool used = mn <= mx;
            const int x0 = used ? mn : 0;
            const int nv = used ? (mx - mn + 1) : 0;
            int first, pad = 0, total;
            {
                const int incl = wave_scan_add(nv);
                first = incl - nv;
                total = __builtin_amdgcn_readlane(incl, 63);
            }
            if (p.row_s >= 0) {
                // Bank-aware row starts (cubic): row r starts at a slot = x0 + r * S (mod 16), i.e. the image behaves like a box
                // with row stride S in the 16-slot bank space of ds_read_b128 while only the spans are stored.  The gaps (< 16
                // vectors per row) are filled from the zero vector.  S comes from the host's model of the gather's lane groups
                // (vt_plan.hip: S = 0 with the service-group lane mapping, quad_row_stride's model otherwise).  If the padded image does
                // not fit the slot, the unpadded prefix sum above stays.
                // Every row start is pinned to a residue c_r = (x0_r + r * S) mod 16, so the gap in front of row r depends on its
                // predecessor alone: gap_r = (c_r - c_{r-1} - n_{r-1}) mod 16 -- the placement is one more prefix sum, not a walk over
                // the rows (round 2 walked them with readlane in a scalar loop: ~35 dependent iterations while three waves waited).
                // Rows in use are usually contiguous (the footprint is convex), but an in-plane minification beyond the stencil's reach
                // leaves unused box rows between them: the predecessor is the last row IN USE before this one (a max-scan of the
                // used rows' indices finds it); the rows before the first one carry residue 0 and length 0.
                const int S = p.row_s;
                const int c_r = (x0 + lane * S) & 15;
                const int end_res = (nv > 0) ? ((c_r + nv) & 15) : 0;            // residue of the position right behind this row
                const int last_used = wave_scan_max((nv > 0) ? lane : -1);
                const int prev_row = wave_shift_up1(last_used, -1);              // last row in use strictly before this one (-1: none)
                int prev_end = __shfl(end_res, max(prev_row, 0));
                if (lane == 0 || prev_row < 0) prev_end = 0;                     // first row in use: the image starts at position 0
                const int gap = (nv > 0) ? ((c_r - prev_end) & 15) : 0;
                const int inc2 = wave_scan_add(gap + nv);
                const int first_p = inc2 - nv, pad_p = gap;
                const int pos = __builtin_amdgcn_readlane(inc2, 63);
                if (((pos + 63) & ~63) * 16 <= p.slot_floats * 4 && pos <= NT * kQuadMaxIt) { first = first_p; pad = pad_p; total = pos; }
            }
            tab[lane] = x0;
            tab[kRowsMax + lane] = first;
            if (lane == 0) tab[2 * kRowsMax] = total;
            const int last = min(first + nv, kVrowCap);
            for (int v = max(first - pad, 0); v < min(first, kVrowCap); ++v) vrow[v] = 255;
            // a row's run of entries: bytes up to the next word, whole words, bytes (runs of different lanes share words only at their ends)
            int v = first;
            for (; v < last && (v & 3); ++v) vrow[v] = (unsigned char)lane;
            const unsigned lane4 = (unsigned)lane * 0x01010101u;
            for (; v + 4 <= last; v += 4) *reinterpret_cast<unsigned*>(vrow + v) = lane4;
            for (; v < last; ++v) vrow[v] = (unsigned char)lane;
        }
        __syncthreads();
#endif
#ifdef VT_EXPERIMENTS
        stamp[3] = clock64();
#endif
        nvec = tab[2 * kRowsMax];
#pragma unroll
        for (int it = 0; it < kQuadMaxIt; ++it) {
            const int v = tid + NT * it;
            const int y = (v < nvec && v < kVrowCap) ? vrow[v] : 255;
            const int yy = (y == 255) ? 0 : y;
            const int cx = v - tab[kRowsMax + yy];
            const int gy = o1 + yy, gx = o2 + tab[yy] + cx;
            const bool ok = (y != 255) && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sW;
            voff[it] = ok ? (gy * p.sPq + 4 * gx) * 4 : p.zero_off_q;
        }
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int row = min(max(iy[px] - HALO + r, 0), kRowsMax - 1);
                q[px][r] = 16 * (tab[kRowsMax + row] + (ix[px] - HALO - tab[row]));
            }
        }
        __syncthreads();                          // the table is dead from here on; the ring may be written
    }
    // voff[] now holds each vector's offset RELATIVE to the zero vector's: a quad outside the resident copy is staged from the zero
    // vector by masking that difference (off = zero_off + (voff & mask), mask wave-uniform) -- pure arithmetic.  The obvious form,
    // `quad_ok ? voff[it] : p.zero_off_q`, selects between two lvalues the nested lambdas below reach through references; the
    // compiler turned it into a select of two ADDRESSES followed by one load, which kept voff[] in scratch memory in the trilinear
    // instantiations: 3 dword stores per thread and workgroup, 9 % more bytes written to HBM than the output itself at 16-plane chunks
    // (WRITE_SIZE 1.088x at 16 planes, 1.044x at 32: profiles/r04_write_size_variants.txt).
#pragma unroll
    for (int it = 0; it < kQuadMaxIt; ++it) voff[it] -= p.zero_off_q;
    const int nvec64 = (nvec + 63) & ~63;         // whole waves stage: a wave's instruction is issued in full or not at all
    const int slot_bytes = p.slot_floats * 4;

    if (nvec64 * 16 > slot_bytes || nvec64 > NT * kQuadMaxIt) {
        // The footprint does not fit the slot planned on the host: gather this workgroup's voxels from the quad copy
        // directly.  Slow, never wrong.  (A separate function with scalar arguments: indexing the per-pixel arrays with a
        // run-time pixel index here would move them to scratch memory for the hot path too.)
#pragma unroll
        for (int px = 0; px < NPIX; ++px)
            if (pix_ok[px])
                quad_gather_column<KIND>(srcq, out, p, (int64_t)(h0 + jh0 + px * RP) * orow + (w0 + kw), d_begin, d_end, in_yx[px],
                                         o1 + iy[px], o2 + ix[px], fy[px], fx[px]);
        return;
    }

    // ---- pipeline ----
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int quad_bytes = p.sH * p.sPq * 4;      // bytes of one resident quad-plane, < 2^31 (host-checked)
    const int nquads_res = (p.sD + 3) >> 2;
    // first / last source plane any output of the chunk taps, and the quads holding them
    const int plane_first = d_begin + p.zoff - ZH, plane_last = d_end - 1 + p.zoff + ZH + ZNEW;
    const int Q0 = floordiv4(plane_first), QN = floordiv4(plane_last);
    const int Q_base = max(0, min(Q0, nquads_res - 1));
    // source: one descriptor for the chunk, based at the first resident quad it touches; the quad is selected with the scalar offset
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(srcq) + (int64_t)Q_base * quad_bytes), 0, 0x7fffffff, 0x00020000);
    // output: based at the tile's first voxel of plane d_begin; per-pixel byte offsets ob[], plane selected with the scalar offset
    __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(out + ((int64_t)d_begin * ostride + (int64_t)h0 * orow + w0)), 0, 0x7fffffff, 0x00020000);
    const int oplane_bytes = (int)(ostride * 4);  // (d_end - d_begin) * oplane_bytes < 2^31 (host-checked)
    const float fz = p.fz;
    float wz[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CUBIC && !ZID) cubic_weights<KIND == 2>(fz, wz);
    // steps whose four outputs all belong to the chunk: quads Qf0 .. Qf1 (output d's newest tap plane is d + zoff + ZH + ZNEW)
    const int tap_new = p.zoff + ZH + ZNEW;
    const int Qf0 = floordiv4(d_begin + tap_new + 3), Qf1 = floordiv4(d_end - 4 + tap_new);
    const bool nt_stores = (p.flags & (1 << 28)) != 0;       // streaming (nontemporal) output stores, chosen by the planner
    const bool tile_fast = all_valid && (h0 + TH <= p.oH) && (w0 + TW <= p.oW);
    char* const lds_c = reinterpret_cast<char*>(lds);
#ifdef VT_EXPERIMENTS      // make EXTRA=-DVT_EXPERIMENTS: VT_EXP_NOSTORE / VT_EXP_NOLOAD / VT_EXP_NOLDS ablations
    const bool no_stores = (p.flags & (1 << 21)) != 0, no_loads = (p.flags & (1 << 22)) != 0, no_lds = (p.flags & (1 << 26)) != 0;
#else
    constexpr bool no_stores = false, no_loads = false, no_lds = false;
#endif
#ifdef VT_EXPERIMENTS
    if (p.flags & (1 << 27)) {                    // VT_EXP_NOLOOP: set-up only; thread 0 leaves the four phase durations of its workgroup
        if (nvec == 12345678) out[0] = (float)(voff[0] + q[0][0]);
        stamp[4] = clock64();
        if (tid == 0) {
            const int64_t b = (int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.y;
            for (int k = 0; k < 4; ++k) out[b * 4 + k] = (float)(stamp[k + 1] - stamp[k]);
        }
        return;
    }
#endif

    auto run = [&](auto nit_c) {
        constexpr int NIT = decltype(nit_c)::value;
        const bool last_wave = wave_first + NT * (NIT - 1) < nvec64;          // wave-uniform
        auto issue_quad = [&](int Q, int slot_off) {
            if (no_loads) return;
            const bool quad_ok = (unsigned)Q < (unsigned)nquads_res;           // wave-uniform
            const int soff = quad_ok ? (Q - Q_base) * quad_bytes : 0;
            const int okmask = quad_ok ? -1 : 0;
            char* dst = lds_c + slot_off + 16 * wave_first;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int off = p.zero_off_q + (voff[it] & okmask);
                if (it + 1 < NIT || last_wave)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (16 * NT) * it), 16, off, soff, 0, 0);
            }
        };

        float carry[NPIX][NC];
#pragma unroll
        for (int px = 0; px < NPIX; ++px)
#pragma unroll
            for (int c = 0; c < NC; ++c) carry[px][c] = 0.f;

        // in-plane partials of the quad's four planes for every pixel of this thread
        auto partials = [&](int slot_off, v4f (&part)[NPIX]) {
            if (no_lds) {
#pragma unroll
                for (int px = 0; px < NPIX; ++px) asm volatile("" : "=v"(part[px]));
                return;
            }
            const char* sl = lds_c + slot_off;
#pragma unroll
            for (int px = 0; px < NPIX; ++px) {
                if constexpr (!CUBIC) {
                    const v4f a00 = *reinterpret_cast<const v4f*>(sl + q[px][0]);
                    const v4f a01 = *reinterpret_cast<const v4f*>(sl + q[px][0] + 16);
                    const v4f a10 = *reinterpret_cast<const v4f*>(sl + q[px][1]);
                    const v4f a11 = *reinterpret_cast<const v4f*>(sl + q[px][1] + 16);
                    const v4f x0 = (a01 - a00) * fx[px] + a00;
                    const v4f x1 = (a11 - a10) * fx[px] + a10;
                    part[px] = (x1 - x0) * fy[px] + x0;
                } else {
                    v4f accy = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const char* rowp = sl + q[px][bb];
                        const v4f t0 = *reinterpret_cast<const v4f*>(rowp);
                        const v4f t1 = *reinterpret_cast<const v4f*>(rowp + 16);
                        const v4f t2 = *reinterpret_cast<const v4f*>(rowp + 32);
                        const v4f t3 = *reinterpret_cast<const v4f*>(rowp + 48);
                        v4f accx = t0 * wx[px][0];
                        accx = t1 * wx[px][1] + accx;
                        accx = t2 * wx[px][2] + accx;
                        accx = t3 * wx[px][3] + accx;
                        accy = accx * wy[px][bb] + accy;
                    }
                    part[px] = accy;
                }
            }
        };
        // axis-0 combination of one new plane partial with the carried ones (same order as the other kernels)
        auto zcombine = [&](int px, float pn) -> float {
            float val;
            if constexpr (ZID) {
                val = pn;                          // == fmaf(0, next - pn, pn) for finite data
            } else if constexpr (!CUBIC) {
                val = fmaf(fz, pn - carry[px][0], carry[px][0]);
                carry[px][0] = pn;
            } else {
                float acc = wz[0] * carry[px][0];
                acc = fmaf(wz[1], carry[px][1], acc);
                acc = fmaf(wz[2], carry[px][2], acc);
                val = fmaf(wz[3], pn, acc);
                carry[px][0] = carry[px][1]; carry[px][1] = carry[px][2]; carry[px][2] = pn;
            }
            return val;
        };

        int slot_off = 0;
        issue_quad(Q0, 0);
        bool prev_full = false;
        for (int Q = Q0; Q <= QN; ++Q) {
            const bool full = tile_fast && Q >= Qf0 && Q <= Qf1;               // wave-uniform
            // Order of this wave's vector-memory operations: loads(Q0) | [loads(Q+1) stores(Q)] ...  When the loads of quad Q are
            // waited for, only the stores of step Q-1 are younger: NSTORE of them after a full step (exact count needed: a
            // smaller number of outstanding operations would satisfy the wait with the loads still in flight).
            if (full && prev_full && !no_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();         // everyone's loads landed; everyone is done reading the other slot
            if (Q < QN) issue_quad(Q + 1, slot_off ^ slot_bytes);
            v4f part[NPIX];
            partials(slot_off, part);
            const int d_first = 4 * Q - tap_new;  // output plane whose newest tap plane is 4Q
            if (full) {
                int soff = (d_first - d_begin) * oplane_bytes;
                float val[4][NPIX];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int px = 0; px < NPIX; ++px) val[i][px] = zcombine(px, part[px][i]);
                // the cache policy of a store is an immediate of the instruction: two copies of the store block
                if (no_stores) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int px = 0; px < NPIX; ++px) asm volatile("" ::"v"(val[i][px]));
                } else if (nt_stores) {
#pragma unroll
                    for (int i = 0; i < 4; ++i, soff += oplane_bytes)
#pragma unroll
                        for (int px = 0; px < NPIX; ++px)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val[i][px]), orsrc, ob[px], soff, 2);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i, soff += oplane_bytes)
#pragma unroll
                        for (int px = 0; px < NPIX; ++px)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val[i][px]), orsrc, ob[px], soff, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = d_first + i;
                    const bool d_ok = (d >= d_begin) && (d < d_end);           // wave-uniform
                    const double ez = (double)d + p.m[3];
                    const bool z_ok = (ez >= p.vlo[0]) && (ez < p.vhi[0]);
                    const int soff = (d - d_begin) * oplane_bytes;
#pragma unroll
                    for (int px = 0; px < NPIX; ++px) {
                        const float val = zcombine(px, part[px][i]);
                        if (d_ok && pix_ok[px]) {
                            if (in_yx[px] && z_ok) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), orsrc, ob[px], soff, 0);
                            else if (!keep) __builtin_amdgcn_raw_buffer_store_b32(0u, orsrc, ob[px], soff, 0);
                        }
                    }
                }
            }
            prev_full = full;
            slot_off ^= slot_bytes;
        }
    };
    using std::integral_constant;
    switch (nvec64 / NT + ((nvec64 % NT) ? 1 : 0)) {
        case 1: run(integral_constant<int, 1>{}); break;
        case 2: run(integral_constant<int, 2>{}); break;
        case 3: run(integral_constant<int, 3>{}); break;
        case 4: run(integral_constant<int, 4>{}); break;
        default: run(integral_constant<int, 5>{}); break;
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
typedef void (*quad_fn)(const float*, float*, const AffineParams);
struct QuadCfg { int th, tw, nt; };
static const QuadCfg kQuad[] = {
    {16, 32, 256},    // 0: two pixels per thread, 128-byte store rows
    {8, 32, 256},     // 1: one pixel per thread, half the footprint (large magnifications of the footprint)
    {32, 32, 256},    // 2: four pixels per thread, square tile (least halo per voxel)
    {16, 64, 256},    // 3: four pixels per thread, 256-byte store rows
    {32, 32, 512},    // 4: two pixels per thread, 8 waves
    {16, 64, 512},    // 5: two pixels per thread, 8 waves, 256-byte store rows
};
int quad_max_it() { return kQuadMaxIt; }
int quad_config_count() { return (int)(sizeof(kQuad) / sizeof(kQuad[0])); }
void quad_config(int idx, int* th, int* tw, int* nt) { *th = kQuad[idx].th; *tw = kQuad[idx].tw; *nt = kQuad[idx].nt; }

template <int TH, int TW, int NT>
static quad_fn pick_quad(int kind)
{
    switch (kind) {
        case 0: return affine_march4<0, TH, TW, NT>;
        case 1: return affine_march4<1, TH, TW, NT>;
        case 3: return affine_march4<3, TH, TW, NT>;
        case 4: return affine_march4<4, TH, TW, NT>;
        default: return affine_march4<2, TH, TW, NT>;
    }
}
static quad_fn quad_entry(int cfg, int kind)
{
    switch (cfg) {
        case 0: return pick_quad<16, 32, 256>(kind);
        case 1: return pick_quad<8, 32, 256>(kind);
        case 2: return pick_quad<32, 32, 256>(kind);
        case 3: return pick_quad<16, 64, 256>(kind);
        case 4: return pick_quad<32, 32, 512>(kind);
        default: return pick_quad<16, 64, 512>(kind);
    }
}

hipError_t init_quad_kernels()
{
    for (int cfg = 0; cfg < quad_config_count(); ++cfg)
        for (int kind = 0; kind < 5; ++kind) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(quad_entry(cfg, kind)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

// Workgroups of this kernel that one CU keeps resident (register- and LDS-limited), from the runtime's occupancy
// calculator; cached per (kernel, LDS size) -- the planner calls this on the per-call path.
int quad_blocks_per_cu(int cfg, int interp, int lds_bytes, bool zid)
{
    static std::mutex mu;
    static std::unordered_map<uint64_t, int> cache;
    const int kind = zid ? (interp_kind(interp) == 0 ? 3 : 4) : interp_kind(interp);
    const uint64_t key = ((uint64_t)cfg << 34) | ((uint64_t)kind << 32) | (uint32_t)lds_bytes;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(quad_entry(cfg, kind)), kQuad[cfg].nt,
                                                     (size_t)lds_bytes) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 1;
    }
    if (std::getenv("VT_DEBUG_ALLOC"))
        std::fprintf(stderr, "[vt] plane-quad kernel kind %d, tile %d x %d, %d threads, %d bytes of LDS: %d workgroups per CU\n", kind, kQuad[cfg].th,
                     kQuad[cfg].tw, kQuad[cfg].nt, lds_bytes, n);
    cache.emplace(key, n);
    return n;
}

hipError_t launch_affine_quad(int cfg, int interp, const float* srcq, float* out, const AffineParams& p,
                              int grid, int lds_bytes, hipStream_t stream)
{
    // integer axis-0 offset (set by plan_quad when fz == 0 exactly): bit 25 = trilinear on one tap plane (KIND 3), bit 19 = cubic on the
    // z-convolved copy (KIND 4; bit 18 selects the `_simple` weight formula)
    const int ik = interp_kind(interp);
    const int kind = ((p.flags & (1 << 25)) && ik == 0) ? 3 : (((p.flags & (1 << 19)) && ik != 0) ? 4 : ik);
    quad_fn fn = quad_entry(cfg, kind);
    const dim3 g = (p.flags & (1 << 29)) ? dim3((unsigned)(p.nTh * p.nTw), (unsigned)p.nTd) : dim3((unsigned)grid);
    hipLaunchKernelGGL(fn, g, dim3(kQuad[cfg].nt), lds_bytes, stream, srcq, out, p);
    return hipGetLastError();
}

}  // namespace vt
