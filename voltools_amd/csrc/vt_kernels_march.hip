// vt_kernels_march.hip -- the round-1 axis-0-separable *marching* transform kernels (gfx950): plain layout (kernel 4) and
// plane-pair layout (kernel 5).  Since round 3 they are compiled only into the test build (`make LEGACY=1`, -DVT_LEGACY): the
// plane-quad kernel (vt_kernels_quad.hip) serves every launch they served, and the planner census (tools/planner_census.py)
// shows them chosen nowhere but for 4x in-plane minification of cubic volumes, which the box kernel now takes.  They stay as
// independent implementations for cross-checks (tests/test_gpu_legacy.py).  The axis-exchange relayout and the footprint-table
// constants that the current kernels share live at the top of this file.
//
//
// Matrices of the block form  [1 0 0 tz; 0 a b ty; 0 c d tx]  (rotations about axis 0 -- the README sweep
// `rotate((0, i, 0))` and every BASELINE configuration --, in-plane scale/shear, any translation): the source plane of an
// output voxel depends only on d, its in-plane position only on (h, w).  The kernel is a software pipeline along axis 0.
//
// A workgroup owns one TH x TW in-plane output tile and marches through `dch` output planes.
//   * Set-up, once per workgroup.  Every thread computes the in-plane tap origin, fractions and weights of its NPIX
//     pixels.  The union of all taps is the tile's *footprint*: a rotated rectangle.  Instead of staging its axis-aligned
//     bounding box (2.7x the tile area at 45 degrees), the workgroup records, per source row, the span [xmin, xmax] that is
//     actually tapped (LDS atomics), aligns each span to 16 bytes, and packs the spans back to back (a 64-entry prefix sum
//     done by one wave).  Each thread then derives, once, the byte offsets of the 16-byte source vectors it will stage for
//     every plane, and the LDS offsets of its taps' rows.  Out-of-volume vectors point at the zero vector the resident
//     layout keeps at the end of every row (= the texture unit's border mode).
//   * Source planes stream through a ring of R = (LA+1)G + 2*HALO + 1 LDS slots filled by `buffer_load ... lds`
//     (direct-to-LDS: one instruction, no VGPR round trip, no VALU work).  While G output planes are computed, the loads
//     of the following LA groups are in flight.
//   * Every source plane's in-plane partial (bilinear blend / 16-tap B-spline sum) is computed once per pixel and carried
//     in registers across the 2 (4) output planes that use it; the z fraction is the same for the whole launch.
//   * One s_barrier per G planes, preceded by a *counted* s_waitcnt vmcnt(N) that leaves later loads and the previous
//     groups' stores in flight.
// Per output voxel: 4 (linear) / 16 (cubic) LDS reads, ~12 / ~30 VALU instructions, one 4-byte store.
// If a tile's packed footprint exceeds the slot size planned on the host (an estimate with margin), that workgroup falls
// back to gathering its voxels straight from global memory -- slower, never wrong.
#include "vt_internal.h"
#include <mutex>
#include <unordered_map>
#include "vt_device.h"
#include "vt_march_common.h"

#include <climits>
#include <type_traits>

namespace vt {

// plain [z][y][P] -> [y][z][P] (axes 0 and 1 exchanged; whole rows incl. the zero pad move as 16-byte vectors)
__global__ __launch_bounds__(256) void relayout_swap01(const float4* __restrict__ src, float4* __restrict__ dst, int D, int H, int P4)
{
    const int xv = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, z = blockIdx.z;
    if (xv >= P4) return;
    dst[((int64_t)y * D + z) * P4 + xv] = src[((int64_t)z * H + y) * P4 + xv];
}

hipError_t launch_relayout_swap01(const float* src, float* dst, int D, int H, int P, hipStream_t stream)
{
    const dim3 grid((P / 4 + 255) / 256, H, D);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(relayout_swap01, grid, dim3(256), 0, stream, reinterpret_cast<const float4*>(src),
                       reinterpret_cast<float4*>(dst), D, H, P / 4);
    return hipGetLastError();
}

int march_rows_max() { return kRowsMax; }
int march_table_bytes() { return kTabBytes; }

#ifdef VT_LEGACY


// in-plane partial of one source plane for one pixel; q[] = LDS float offsets of the tap rows inside the plane slot
template <int KIND, int NR>
__device__ __forceinline__ float plane_partial_rows(const float* __restrict__ pl, const int (&q)[NR], float fy, float fx,
                                                    const float (&wy)[4], const float (&wx)[4])
{
    if constexpr (KIND == 0) {
        const float* r0 = pl + q[0];
        const float* r1 = pl + q[1];
        const float a00 = r0[0], a01 = r0[1], a10 = r1[0], a11 = r1[1];
        const float x0 = fmaf(fx, a01 - a00, a00);
        const float x1 = fmaf(fx, a11 - a10, a10);
        return fmaf(fy, x1 - x0, x0);
    } else {
        float accy = 0.f;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float* rowp = pl + q[bb];
            float accx = wx[0] * rowp[0];
            accx = fmaf(wx[1], rowp[1], accx);
            accx = fmaf(wx[2], rowp[2], accx);
            accx = fmaf(wx[3], rowp[3], accx);
            accy = fmaf(wy[bb], accx, accy);
        }
        return accy;
    }
}

// Register budget of the pair kernel.  Natural allocation: 138 VGPRs -> 3 waves per SIMD.  Forcing 4 waves (128 VGPRs,
// 10 spilled) was measured on one box: same time at 512^3, +1-2 % at 1024^3, but 25 % more HBM reads at 20 degrees (more
// workgroups in flight than the XCD's L2 can keep the halos for) -- so the natural allocation stays.  -DVT_ZPAIR_WAVES=4 to retry.
#ifndef VT_ZPAIR_WAVES
#define VT_ZPAIR_WAVES 0
#endif
#if VT_ZPAIR_WAVES > 0
#define VT_ZPAIR_OCC __attribute__((amdgpu_waves_per_eu(VT_ZPAIR_WAVES, VT_ZPAIR_WAVES)))
#else
#define VT_ZPAIR_OCC
#endif

template <int KIND, int TH, int TW, int G, int LA, int NT>
__global__ __launch_bounds__(NT) void affine_march_zsep(const float* __restrict__ src, float* __restrict__ out,
                                                          const AffineParams p)
{
    static_assert(NT % TW == 0 && TH % (NT / TW) == 0 && NT % 64 == 0, "tile/thread mapping");
    constexpr bool CUBIC = KIND != 0;
    constexpr int HALO = CUBIC ? 1 : 0;
    constexpr int NR = 2 + 2 * HALO;              // tap rows (and columns) per pixel
    constexpr int NC = 2 * HALO + 1;              // carried partials per pixel
    // ring slots: live group + `la` groups in flight (+ halo planes); la >= LA is chosen on the host (p.Lz slots)
    const int R = p.Lz;
    const int la = (R - 2 * HALO - 1) / G - 1;
    constexpr int RP = NT / TW;
    constexpr int NPIX = TH / RP;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    int tw_i, th_i, chunk;
    march_tile(p, t, th_i, tw_i, chunk);
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int d_begin = chunk * p.dch;
    const int d_end = min(d_begin + p.dch, p.oD);

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
    // every lane of the workgroup stores exactly one value per pixel and plane -> the number of stores a wave has in
    // flight is known, and the wait before the barrier can leave them (and later loads) outstanding
    const int64_t ostride = p.ostride, orow = p.orow;     // element strides of an output plane / row (axis swaps)
    const int kw = tid % TW;
    const int jh0 = tid / TW;

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

    const int o1 = (int)floor(lo[1]) - HALO;
    const int o2 = ((int)floor(lo[2]) - HALO) & ~3;
    const int Ly = p.Ly;                          // <= kRowsMax

    // ---- per-pixel tap geometry ----
    int iy[NPIX], ix[NPIX];
    float fy[NPIX], fx[NPIX];
    float wy[NPIX][4], wx[NPIX][4];
    bool in_yx[NPIX];
    int64_t ooff[NPIX];                           // element offset of the pixel in output plane d_begin
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
        if constexpr (CUBIC) { cubic_weights<KIND == 2>(fy[px], wy[px]); cubic_weights<KIND == 2>(fx[px], wx[px]); }
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { wy[px][k] = 0.f; wx[px][k] = 0.f; }
        }
        // rows 1, 2 of an axis-0-separable matrix ignore d
        in_yx[px] = all_valid || (canonical_inside_axis(p, 1, 0, h0 + j, w0 + kw) && canonical_inside_axis(p, 2, 0, h0 + j, w0 + kw));
        ooff[px] = (int64_t)d_begin * ostride + (int64_t)(h0 + j) * orow + (w0 + kw);
    }

    int voff[kMaxIt];                             // byte offset inside a source plane of each 16-byte vector this thread stages
    int q[NPIX][NR];                              // float offset of each tap row of each pixel inside a plane slot
    int nvec;                                     // 16-byte vectors per plane slot
    const int slot_floats = p.slot_floats;
    const bool box_mode = (p.flags & (1 << 20)) != 0;
    if (box_mode) {
        // ---- bounding box, row stride Lx (chosen on the host for few bank conflicts) ----
        const int Lx = p.Lx, nvx = Lx >> 2;
        nvec = Ly * nvx;
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            const int v = tid + NT * it;
            const int y = v / nvx;
            const int cx = v - y * nvx;
            const int gy = o1 + y, gx = o2 + 4 * cx;
            // stride-padding columns are never tapped: fetch the zero vector for them instead of source data
            const bool ok = (v < nvec) && 4 * cx < p.Lx_used && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sP;
            voff[it] = ok ? (gy * p.sP + gx) * 4 : p.zero_off;
        }
#pragma unroll
        for (int px = 0; px < NPIX; ++px)
#pragma unroll
            for (int r = 0; r < NR; ++r) q[px][r] = __mul24(iy[px] - HALO + r, Lx) + (ix[px] - HALO);
    } else {
        // ---- row spans of the footprint, packed ----
        // tab[0..63]: per-row xmin -> x0 (aligned span start); tab[64..127]: per-row xmax -> first vector of the row;
        // tab[128]: total vectors.  The table overlays the ring (not in use yet).
        int* tab = reinterpret_cast<int*>(lds);
        build_span_table<TH, TW, HALO, 4>(tab, p, Ly, by, bx, tid);
        const unsigned char* vrow = reinterpret_cast<const unsigned char*>(tab + kTabInts);
        nvec = tab[2 * kRowsMax];
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            const int v = tid + NT * it;
            const int y = (v < nvec) ? vrow[v] : 0;
            const int cx = v - tab[kRowsMax + y];
            const int gy = o1 + y, gx = o2 + tab[y] + 4 * cx;
            const bool ok = (v < nvec) && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sP;
            voff[it] = ok ? (gy * p.sP + gx) * 4 : p.zero_off;
        }
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int row = min(max(iy[px] - HALO + r, 0), kRowsMax - 1);
                q[px][r] = 4 * tab[kRowsMax + row] + (ix[px] - HALO - tab[row]);
            }
        }
        __syncthreads();                          // the table is dead from here on; the ring may be written
    }

    if (nvec * 4 > slot_floats || nvec > NT * kMaxIt) {
        // The footprint does not fit the slot planned on the host: gather this workgroup's voxels from global memory
        // (same arithmetic as affine_direct).
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
            const int h = h0 + jh0 + px * RP, w = w0 + kw;
            if (h >= p.oH || w >= p.oW) continue;
            int64_t oo = ooff[px];
            for (int d = d_begin; d < d_end; ++d, oo += ostride) {
                const double ez = (double)d + p.m[3];
                const bool inside = in_yx[px] && (ez >= p.vlo[0]) && (ez < p.vhi[0]);
                if (inside) out[oo] = direct_sample<KIND>(src, p, d + p.zoff, o1 + iy[px], o2 + ix[px], p.fz, fy[px], fx[px]);
                else if (!keep) out[oo] = 0.0f;
            }
        }
        return;
    }

    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int plane_bytes = p.sH * p.sP * 4;      // < 2^31 (host-checked)
    // one buffer descriptor for the whole chunk, based at the first resident plane it touches; the plane is selected with
    // the scalar offset operand
    const int P_first = d_begin + p.zoff - HALO;
    const int P_base = max(0, min(P_first, p.sD - 1));
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(src) + (int64_t)P_base * plane_bytes), 0, 0x7fffffff, 0x00020000);
    const float fz = p.fz;
    float wz[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CUBIC) cubic_weights<KIND == 2>(fz, wz);
    const int ngroups = (d_end - d_begin + G - 1) / G;
#ifdef VT_EXPERIMENTS      // make EXTRA=-DVT_EXPERIMENTS: VT_EXP_NOSTORE / VT_EXP_NOLOAD ablations
    const bool no_stores = (p.flags & (1 << 21)) != 0, no_loads = (p.flags & (1 << 22)) != 0;
#else
    constexpr bool no_stores = false, no_loads = false;
#endif

    // ---- pipeline ----
    // The CU has ONE scalar unit: the steady-state loop is written so that it needs a few dozen scalar instructions per
    // group instead of a few hundred (measured: with neither loads nor stores the first version still took 62 % of its
    // run time, 70 % of that on the scalar unit).  Hence: the number of staging instructions per plane (NIT) is a
    // compile-time constant selected by one switch, only the last of them is lane-masked, and the common case -- tile
    // fully inside the valid region and the output -- has no per-voxel predicates at all.
    // Order of a wave's vector-memory operations: loads(0) | [loads(g+1) stores(g)] for g = 0, 1, ...  => when group g is
    // about to be computed only the stores of group g-1 were issued after its loads.
    auto run = [&](auto nit_c, auto fast_c) {
        constexpr int NIT = decltype(nit_c)::value;
        constexpr bool FAST = decltype(fast_c)::value;
        const bool last_ok = tid + NT * (NIT - 1) < nvec;
        auto issue_plane = [&](int P, int slot) {
            if (no_loads) return;
            const bool plane_ok = (unsigned)P < (unsigned)p.sD;                  // wave-uniform
            float* dst = lds + slot * slot_floats + 4 * wave_first;
            if (plane_ok) {
                const int soff = (P - P_base) * plane_bytes;
#pragma unroll
                for (int it = 0; it < NIT - 1; ++it)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (4 * NT) * it), 16, voff[it], soff, 0, 0);
                if (last_ok)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (4 * NT) * (NIT - 1)), 16, voff[NIT - 1], soff, 0, 0);
            } else {                                                             // plane outside the volume: border zeros
#pragma unroll
                for (int it = 0; it < NIT - 1; ++it)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (4 * NT) * it), 16, p.zero_off, 0, 0, 0);
                if (last_ok)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (4 * NT) * (NIT - 1)), 16, p.zero_off, 0, 0, 0);
            }
        };
        int P_next = P_first;                     // next source plane to stage
        int slot_next = 0;
        for (int c = 0; c < G + 2 * HALO + 1; ++c) {          // group 0 with its halo planes
            issue_plane(P_next, slot_next);
            ++P_next;
            slot_next = (slot_next + 1 == R) ? 0 : slot_next + 1;
        }
        for (int a = 1; a < la && a < ngroups; ++a) {         // groups 1 .. la-1
#pragma unroll
            for (int c = 0; c < G; ++c) {
                issue_plane(P_next, slot_next);
                ++P_next;
                slot_next = (slot_next + 1 == R) ? 0 : slot_next + 1;
            }
        }
        // staging instructions this wave issues per plane: NIT, or NIT-1 when none of its lanes is in the last one
        const bool wave_has_last = __builtin_amdgcn_readfirstlane(tid & ~63) + NT * (NIT - 1) < nvec;
        const int nload = G * (wave_has_last ? NIT : NIT - 1);    // vector-memory instructions per staged group
        constexpr int nstore = G * NPIX;                          // ... and per stored group (FAST: never predicated)
        float carry[NPIX][NC];
        int slot_cur = 0;                         // slot of source plane zs(d) - HALO
        int64_t dofs = 0;                         // element offset of output plane d relative to d_begin (wave-uniform)
        for (int g = 0; g < ngroups; ++g, dofs += (int64_t)G * ostride) {
            // Iteration j issues [loads(j+la), stores(j)], so after loads(g) this wave has issued stores(max(0,g-la) .. g-1)
            // and loads(g+1 .. min(g+la-1, ngroups-1)): wait until only those are outstanding.
            if (FAST && !no_stores && !no_loads) {
                if (la == 1) {                    // the default depth: two cases, immediates (no jump table in the hot loop)
                    if (g > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nstore) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    const int n = min(g, la) * nstore + min(la - 1, ngroups - 1 - g) * nload;
                    wait_vmcnt_le(n);
                }
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();         // everyone's loads landed; everyone is done with the slots reused next
            if (g + la < ngroups) {
#pragma unroll
                for (int c = 0; c < G; ++c) {
                    issue_plane(P_next, slot_next);
                    ++P_next;
                    slot_next = (slot_next + 1 == R) ? 0 : slot_next + 1;
                }
            }
            if (g == 0) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float* pl = lds + c * slot_floats;
#pragma unroll
                    for (int px = 0; px < NPIX; ++px)
                        carry[px][c] = plane_partial_rows<KIND, NR>(pl, q[px], fy[px], fx[px], wy[px], wx[px]);
                }
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                int sl = slot_cur + NC + i;
                sl = (sl >= R) ? sl - R : sl;
                const float* pl = lds + sl * slot_floats;
                float val[NPIX];
#pragma unroll
                for (int px = 0; px < NPIX; ++px) {
                    const float pn = plane_partial_rows<KIND, NR>(pl, q[px], fy[px], fx[px], wy[px], wx[px]);
                    if constexpr (!CUBIC) {
                        val[px] = fmaf(fz, pn - carry[px][0], carry[px][0]);
                        carry[px][0] = pn;
                    } else {
                        float acc = wz[0] * carry[px][0];
                        acc = fmaf(wz[1], carry[px][1], acc);
                        acc = fmaf(wz[2], carry[px][2], acc);
                        val[px] = fmaf(wz[3], pn, acc);
                        carry[px][0] = carry[px][1]; carry[px][1] = carry[px][2]; carry[px][2] = pn;
                    }
                }
                if constexpr (FAST) {
                    if (!no_stores) {
#pragma unroll
                        for (int px = 0; px < NPIX; ++px) __builtin_nontemporal_store(val[px], &out[ooff[px] + dofs + i * ostride]);
                    } else if (val[0] == 12345.678f) out[0] = val[0];           // keep the values alive
                } else {
                    const int d = d_begin + g * G + i;
                    const double ez = (double)d + p.m[3];
                    const bool z_ok = (ez >= p.vlo[0]) && (ez < p.vhi[0]);
                    if (d < d_end) {
#pragma unroll
                        for (int px = 0; px < NPIX; ++px) {
                            if (h0 + jh0 + px * RP < p.oH && w0 + kw < p.oW) {
                                if (in_yx[px] && z_ok) out[ooff[px] + dofs + i * ostride] = val[px];
                                else if (!keep) out[ooff[px] + dofs + i * ostride] = 0.0f;
                            }
                        }
                    }
                }
            }
            slot_cur += G;
            slot_cur = (slot_cur >= R) ? slot_cur - R : slot_cur;
        }
    };
    const int nit = (nvec + NT - 1) / NT;         // 1 .. kMaxIt
    const bool fast = all_valid && (h0 + TH <= p.oH) && (w0 + TW <= p.oW) && ((d_end - d_begin) % G == 0);
    using std::integral_constant;
    if (fast) {
        switch (nit) {
            case 1: run(integral_constant<int, 1>{}, integral_constant<bool, true>{}); break;
            case 2: run(integral_constant<int, 2>{}, integral_constant<bool, true>{}); break;
            case 3: run(integral_constant<int, 3>{}, integral_constant<bool, true>{}); break;
            default: run(integral_constant<int, 4>{}, integral_constant<bool, true>{}); break;
        }
    } else {
        switch (nit) {
            case 1: run(integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
            case 2: run(integral_constant<int, 2>{}, integral_constant<bool, false>{}); break;
            case 3: run(integral_constant<int, 3>{}, integral_constant<bool, false>{}); break;
            default: run(integral_constant<int, 4>{}, integral_constant<bool, false>{}); break;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// cubic marching kernel on the plane-pair layout
// ---------------------------------------------------------------------------------------------------
// The cubic marching kernel above is bound by LDS reads: 16 dwords per voxel with `ds_read2_b32` (128 B/clk/CU).
// Here the resident source is a second copy in which planes 2p and 2p+1 are interleaved element by element
// ([p][y][x][2], built once by relayout_zpair).  One `ds_read_b64` (256 B/clk/CU) then delivers a tap for two source planes,
// the in-plane 16-tap sums of both planes are formed with packed-f32 FMAs, and the ring shrinks to LA+1 pair slots because
// every plane's partial is consumed exactly once (the z history lives in registers).
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND, int TH, int TW, int LA, int NT>
__global__ __launch_bounds__(NT) VT_ZPAIR_OCC void affine_march_zpair(const float* __restrict__ src2, float* __restrict__ out,
                                                           const AffineParams p)
{
    static_assert(KIND != 0, "cubic only");
    static_assert(NT % TW == 0 && TH % (NT / TW) == 0 && NT % 64 == 0, "tile/thread mapping");
    constexpr int HALO = 1;
    // ring slots (plane pairs): the live pair + `la` pairs in flight; la (1..3) is chosen on the host (p.Lz slots)
    const int R = p.Lz;
    const int la = min(max(R - 1, 1), 3);
    constexpr int RP = NT / TW;
    constexpr int NPIX = TH / RP;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    int tw_i, th_i, chunk;
    march_tile(p, t, th_i, tw_i, chunk);
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int d_begin = chunk * p.dch;
    const int d_end = min(d_begin + p.dch, p.oD);

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
    const bool exact_stores = (h0 + TH <= p.oH) && (w0 + TW <= p.oW) && (all_valid || !keep);
    const int64_t ostride = p.ostride, orow = p.orow;
    const int kw = tid % TW;
    const int jh0 = tid / TW;

    if (!any_valid) {
        if (!keep) {
#pragma unroll
            for (int px = 0; px < NPIX; ++px) {
                const int h = h0 + jh0 + px * RP, w = w0 + kw;
                if (h < p.oH && w < p.oW) {
                    int64_t oo = (int64_t)d_begin * ostride + (int64_t)h * orow + w;
                    for (int d = d_begin; d < d_end; ++d, oo += ostride) out[oo] = 0.0f;
                }
            }
        }
        return;
    }

    const int o1 = (int)floor(lo[1]) - HALO;
    const int o2 = ((int)floor(lo[2]) - HALO) & ~1;            // 16-byte vectors hold 2 positions x 2 planes
    const int Ly = p.Ly, Lx = p.Lx;                              // Lx in positions, even
    const int slot_floats = p.slot_floats;
    const bool box_mode = (p.flags & (1 << 20)) != 0;
    const double by = base[1] - (double)o1, bx = base[2] - (double)o2;

    int iy[NPIX], ix[NPIX];
    float wy[NPIX][4], wx[NPIX][4];
    bool in_yx[NPIX];
    int64_t ooff[NPIX];
#pragma unroll
    for (int px = 0; px < NPIX; ++px) {
        const int j = jh0 + px * RP;
        const double sy = fma(p.m[5], (double)j, fma(p.m[6], (double)kw, by));
        const double sx = fma(p.m[9], (double)j, fma(p.m[10], (double)kw, bx));
        const double fyd = floor(sy), fxd = floor(sx);
        cubic_weights<KIND == 2>((float)(sy - fyd), wy[px]);
        cubic_weights<KIND == 2>((float)(sx - fxd), wx[px]);
        iy[px] = (int)fyd;
        ix[px] = (int)fxd;
        // rows 1, 2 of an axis-0-separable matrix ignore d
        in_yx[px] = all_valid || (canonical_inside_axis(p, 1, 0, h0 + j, w0 + kw) && canonical_inside_axis(p, 2, 0, h0 + j, w0 + kw));
        ooff[px] = (int64_t)(h0 + j) * orow + (w0 + kw);
    }

    int voff[kMaxIt];
    int q[NPIX][4];                               // float offset of tap (iy-1+bb, ix-1) inside a pair slot
    int nvec;
    if (box_mode) {
        const int nvx = Lx >> 1;
        nvec = Ly * nvx;
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            const int v = tid + NT * it;
            const int y = v / nvx;
            const int cx = v - y * nvx;
            const int gy = o1 + y, gx = o2 + 2 * cx;
            const bool ok = (v < nvec) && 2 * cx < p.Lx_used && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)(p.sP2 >> 1);
            voff[it] = ok ? (gy * p.sP2 + 2 * gx) * 4 : p.zero_off2;
        }
#pragma unroll
        for (int px = 0; px < NPIX; ++px)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) q[px][bb] = 2 * (__mul24(iy[px] - HALO + bb, Lx) + (ix[px] - HALO));
    } else {
        // packed row spans (see affine_march_zsep); spans are aligned to 2 positions = one 16-byte vector
        int* tab = reinterpret_cast<int*>(lds);
        build_span_table<TH, TW, HALO, 2>(tab, p, Ly, by, bx, tid);
        const unsigned char* vrow = reinterpret_cast<const unsigned char*>(tab + kTabInts);
        nvec = tab[2 * kRowsMax];
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            const int v = tid + NT * it;
            const int y = (v < nvec) ? vrow[v] : 0;
            const int cx = v - tab[kRowsMax + y];
            const int gy = o1 + y, gx = o2 + tab[y] + 2 * cx;
            const bool ok = (v < nvec) && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)(p.sP2 >> 1);
            voff[it] = ok ? (gy * p.sP2 + 2 * gx) * 4 : p.zero_off2;
        }
#pragma unroll
        for (int px = 0; px < NPIX; ++px)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int row = min(max(iy[px] - HALO + bb, 0), kRowsMax - 1);
                q[px][bb] = 4 * tab[kRowsMax + row] + 2 * (ix[px] - HALO - tab[row]);
            }
        __syncthreads();
    }
    if (nvec * 4 > slot_floats || nvec > NT * kMaxIt) {
        // footprint larger than planned: gather from the plain layout is not available here (src2 is the pair copy), so
        // read the pair copy directly -- element (z, y, x) sits at ((z>>1)*H + y)*P2 + 2x + (z&1)
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
            const int h = h0 + jh0 + px * RP, w = w0 + kw;
            if (h >= p.oH || w >= p.oW) continue;
            float wzl[4];
            cubic_weights<KIND == 2>(p.fz, wzl);
            for (int d = d_begin; d < d_end; ++d) {
                const double ez = (double)d + p.m[3];
                const bool inside = in_yx[px] && (ez >= p.vlo[0]) && (ez < p.vhi[0]);
                float val = 0.f;
                if (inside) {
                    for (int c = 0; c < 4; ++c) {
                        const int z = d + p.zoff - 1 + c;
                        float accy = 0.f;
                        for (int bb = 0; bb < 4; ++bb) {
                            const int y = o1 + iy[px] - 1 + bb;
                            float accx = 0.f;
                            for (int a2 = 0; a2 < 4; ++a2) {
                                const int x = o2 + ix[px] - 1 + a2;
                                float tv = 0.f;
                                if ((unsigned)z < (unsigned)p.sD && (unsigned)y < (unsigned)p.sH && (unsigned)x < (unsigned)p.sW)
                                    tv = src2[((int64_t)(z >> 1) * p.sH + y) * p.sP2 + 2 * x + (z & 1)];
                                accx = (a2 == 0) ? wx[px][0] * tv : fmaf(wx[px][a2], tv, accx);
                            }
                            accy = fmaf(wy[px][bb], accx, accy);
                        }
                        val = fmaf(wzl[c], accy, val);
                    }
                }
                if (inside) out[ooff[px] + (int64_t)d * ostride] = val;
                else if (!keep) out[ooff[px] + (int64_t)d * ostride] = 0.0f;
            }
        }
        return;
    }

    float wz[4];
    cubic_weights<KIND == 2>(p.fz, wz);

    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int pair_bytes = p.sH * p.sP2 * 4;      // < 2^31 (host-checked)
    const int npairs_res = (p.sD + 1) >> 1;       // resident pairs
    // first / last source plane any output of the chunk taps, and the pairs holding them
    const int plane_first = d_begin + p.zoff - HALO, plane_last = d_end - 1 + p.zoff + 2;
    const int Pp0 = plane_first >> 1, PpN = plane_last >> 1;       // arithmetic shift = floor division
    const int Pp_base = max(0, min(Pp0, npairs_res - 1));
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(src2) + (int64_t)Pp_base * pair_bytes), 0, 0x7fffffff, 0x00020000);

    // Specialised on NIT = staging vectors per thread (0 = run-time count with per-vector predicates) and FAST = every
    // output voxel inside the source and the tile fully inside the output: the steady-state loop of the common case then
    // carries no per-vector or per-pixel predicates (they cost scalar exec-mask work every pair).
#ifdef VT_EXPERIMENTS      // make EXTRA=-DVT_EXPERIMENTS: VT_EXP_NOSTORE / VT_EXP_NOLOAD / VT_EXP_NOLDS ablations
    const bool no_stores = (p.flags & (1 << 21)) != 0, no_loads = (p.flags & (1 << 22)) != 0, no_lds = (p.flags & (1 << 26)) != 0;
#else
    constexpr bool no_stores = false, no_loads = false, no_lds = false;
#endif
    auto run = [&](auto nit_c, auto fast_c) {
    constexpr int NIT = decltype(nit_c)::value;
    constexpr bool FAST = decltype(fast_c)::value;
    constexpr int NLOOP = NIT > 0 ? NIT : kMaxIt;
    const bool last_lane = tid + NT * (NLOOP - 1) < nvec;
    const bool last_wave = wave_first + NT * (NLOOP - 1) < nvec;   // wave-uniform
    auto issue_pair = [&](int Pp, int slot) {
        if (no_loads) return;
        const bool pair_ok = (unsigned)Pp < (unsigned)npairs_res;
        const int soff = pair_ok ? (Pp - Pp_base) * pair_bytes : 0;
        float* dst = lds + slot * slot_floats + 4 * wave_first;
#pragma unroll
        for (int it = 0; it < NLOOP; ++it) {
            const int off = pair_ok ? voff[it] : p.zero_off2;
            if (NIT > 0 && it + 1 < NIT) {                   // full iterations: every lane stages a vector
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (4 * NT) * it),
                                                         16, off, soff, 0, 0);
            } else if (NIT > 0) {
                if (last_wave) {
                    if (last_lane)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (4 * NT) * it),
                                                                 16, off, soff, 0, 0);
                }
            } else if (wave_first + NT * it < nvec) {        // wave-uniform
                if (tid + NT * it < nvec)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (4 * NT) * it),
                                                             16, off, soff, 0, 0);
            }
        }
    };

    int nload_w = 0;                              // staging instructions this wave issues per pair (wave-uniform)
    if (NIT > 0) {
        nload_w = NIT - 1 + (last_wave ? 1 : 0);
    } else {
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) nload_w += (wave_first + NT * it < nvec) ? 1 : 0;
    }
    const int npairs = PpN - Pp0 + 1;
    int Pp_next = Pp0, slot_next = 0;
    for (int a = 0; a < la && Pp_next <= PpN; ++a) {
        issue_pair(Pp_next, slot_next);
        ++Pp_next;
        slot_next = (slot_next + 1 == R) ? 0 : slot_next + 1;
    }
    // in-plane partials of the three previous source planes.  (Tried: the z sums of a pair's two outputs as five packed FMAs
    // instead of eight scalar ones, same summation order -- no measurable difference, the loop is not VALU-bound.)
    float c0[NPIX], c1[NPIX], c2[NPIX];
#pragma unroll
    for (int px = 0; px < NPIX; ++px) { c0[px] = 0.f; c1[px] = 0.f; c2[px] = 0.f; }
    int slot_cur = 0;
    int st_hist[3] = {0, 0, 0};                   // store instructions this wave issued in the last three iterations
    int stores_prev = 0;
    for (int Pp = Pp0; Pp <= PpN; ++Pp) {
        // Iteration j issues [loads(j + la), stores(j)]: after this pair's loads the wave has issued the stores of the last
        // `la` iterations and the loads of the next la-1 pairs; wait until only those are outstanding.
        if (FAST || exact_stores) {
            if (la == 1) {                        // the default depth: three cases, immediates (no jump table in the hot loop)
                if (st_hist[0] == 2 * NPIX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPIX) : "memory");
                else if (st_hist[0] == NPIX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIX) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (NIT > 0 && la == 2 && st_hist[0] == 2 * NPIX && st_hist[1] == 2 * NPIX && Pp < PpN) {
                // steady state of depth 2: the stores of two iterations and one pair of loads stay in flight
                if (last_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NPIX + NLOOP) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NPIX + NLOOP - 1) : "memory");
            } else {
                const int i = Pp - Pp0;
                int n = st_hist[0] + (la > 1 ? st_hist[1] : 0) + (la > 2 ? st_hist[2] : 0);
                n += min(la - 1, npairs - 1 - i) * nload_w;
                wait_vmcnt_le(n);
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (Pp_next <= PpN) {
            issue_pair(Pp_next, slot_next);
            ++Pp_next;
            slot_next = (slot_next + 1 == R) ? 0 : slot_next + 1;
        }
        const float* pl = lds + slot_cur * slot_floats;
        // 16 taps x 2 planes per pixel as sixteen ds_read_b64 (2 LDS cycles each).  Written as inline asm because hipcc
        // fuses adjacent 8-byte LDS loads into ds_read2_b64, which runs at half the bytes per cycle (8 cycles per pair).
        v2f part[NPIX];
        const unsigned pl_addr = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)pl;
        v2f tap[NPIX][16];
        if (no_lds) {
#pragma unroll
            for (int px = 0; px < NPIX; ++px)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("" : "=v"(tap[px][k]));
        } else
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const unsigned addr = pl_addr + 4u * (unsigned)q[px][bb];
                asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24"
                             : "=&v"(tap[px][4 * bb]), "=&v"(tap[px][4 * bb + 1]), "=&v"(tap[px][4 * bb + 2]), "=&v"(tap[px][4 * bb + 3])
                             : "v"(addr));
            }
        }
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
            // in-order return: everything but the (NPIX-1-px)*16 youngest reads has landed
            if (px + 1 < NPIX)
                asm volatile("s_waitcnt lgkmcnt(%16)"
                             : "+v"(tap[px][0]), "+v"(tap[px][1]), "+v"(tap[px][2]), "+v"(tap[px][3]), "+v"(tap[px][4]), "+v"(tap[px][5]),
                               "+v"(tap[px][6]), "+v"(tap[px][7]), "+v"(tap[px][8]), "+v"(tap[px][9]), "+v"(tap[px][10]), "+v"(tap[px][11]),
                               "+v"(tap[px][12]), "+v"(tap[px][13]), "+v"(tap[px][14]), "+v"(tap[px][15])
                             : "n"((NPIX - 1 - px) * 16 > 15 ? 15 : (NPIX - 1 - px) * 16));
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(tap[px][0]), "+v"(tap[px][1]), "+v"(tap[px][2]), "+v"(tap[px][3]), "+v"(tap[px][4]), "+v"(tap[px][5]),
                               "+v"(tap[px][6]), "+v"(tap[px][7]), "+v"(tap[px][8]), "+v"(tap[px][9]), "+v"(tap[px][10]), "+v"(tap[px][11]),
                               "+v"(tap[px][12]), "+v"(tap[px][13]), "+v"(tap[px][14]), "+v"(tap[px][15]));
            v2f accy = {0.f, 0.f};
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                v2f accx = tap[px][4 * bb] * wx[px][0];
                accx = tap[px][4 * bb + 1] * wx[px][1] + accx;
                accx = tap[px][4 * bb + 2] * wx[px][2] + accx;
                accx = tap[px][4 * bb + 3] * wx[px][3] + accx;
                accy = accx * wy[px][bb] + accy;
            }
            part[px] = accy;
        }
        stores_prev = 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int plane = 2 * Pp + half;      // newest tap plane of output d
            const int d = plane - 2 - p.zoff;
            const bool d_ok = (d >= d_begin) && (d < d_end);          // wave-uniform
            bool z_ok = true;
            if (!FAST && !all_valid) {
                const double ez = (double)d + p.m[3];
                z_ok = (ez >= p.vlo[0]) && (ez < p.vhi[0]);
            }
            float val[NPIX];
#pragma unroll
            for (int px = 0; px < NPIX; ++px) {
                const float pn = half ? part[px].y : part[px].x;
                float acc = wz[0] * c0[px];
                acc = fmaf(wz[1], c1[px], acc);
                acc = fmaf(wz[2], c2[px], acc);
                val[px] = fmaf(wz[3], pn, acc);
                c0[px] = c1[px]; c1[px] = c2[px]; c2[px] = pn;
            }
            if (d_ok) {
                const int64_t dofs = (int64_t)d * ostride;
                if (FAST && no_stores) {
#pragma unroll
                    for (int px = 0; px < NPIX; ++px) asm volatile("" ::"v"(val[px]));
                } else if (FAST) {
#pragma unroll
                    // streaming stores: the output is written once and not read again; keeping it out of the L2's way is worth
                    // 2-3 % ([measured] 512^3 0.251 -> 0.244 ms at 0 degrees, 0.294 -> 0.288 at 30; 1024^3 2.149 -> 2.102)
                    for (int px = 0; px < NPIX; ++px) __builtin_nontemporal_store(val[px], &out[ooff[px] + dofs]);
                    stores_prev += NPIX;
                } else if (exact_stores) {
#pragma unroll
                    for (int px = 0; px < NPIX; ++px) __builtin_nontemporal_store((in_yx[px] && z_ok) ? val[px] : 0.0f, &out[ooff[px] + dofs]);
                    stores_prev += NPIX;
                } else {
#pragma unroll
                    for (int px = 0; px < NPIX; ++px) {
                        if (h0 + jh0 + px * RP < p.oH && w0 + kw < p.oW) {
                            if (in_yx[px] && z_ok) out[ooff[px] + dofs] = val[px];
                            else if (!keep) out[ooff[px] + dofs] = 0.0f;
                        }
                    }
                }
            }
        }
        st_hist[2] = st_hist[1]; st_hist[1] = st_hist[0]; st_hist[0] = stores_prev;
        slot_cur = (slot_cur + 1 == R) ? 0 : slot_cur + 1;
    }
    };
    using std::integral_constant;
    if (all_valid && exact_stores) {
        switch ((nvec + NT - 1) / NT) {
            case 1: run(integral_constant<int, 1>{}, integral_constant<bool, true>{}); break;
            case 2: run(integral_constant<int, 2>{}, integral_constant<bool, true>{}); break;
            case 3: run(integral_constant<int, 3>{}, integral_constant<bool, true>{}); break;
            default: run(integral_constant<int, 4>{}, integral_constant<bool, true>{}); break;
        }
    } else {
        run(integral_constant<int, 0>{}, integral_constant<bool, false>{});
    }
}

// plain [z][y][P] -> pair layout [z/2][y][P2] with element (z, y, x) at 2x + (z & 1); pad columns stay zero
__global__ __launch_bounds__(256) void relayout_zpair(const float* __restrict__ src, float* __restrict__ dst,
                                                       int D, int H, int W, int P, int P2)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int pp = blockIdx.z;
    if (x >= W) return;
    const int z0 = 2 * pp, z1 = z0 + 1;
    const float a = src[((int64_t)z0 * H + y) * P + x];
    const float b = (z1 < D) ? src[((int64_t)z1 * H + y) * P + x] : 0.0f;
    float2* o = reinterpret_cast<float2*>(dst + ((int64_t)pp * H + y) * P2) + x;
    *o = make_float2(a, b);
}

hipError_t launch_relayout_zpair(const float* src, float* dst, int D, int H, int W, int P, int P2, hipStream_t stream)
{
    const dim3 grid((W + 255) / 256, H, (D + 1) / 2);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(relayout_zpair, grid, dim3(256), 0, stream, src, dst, D, H, W, P, P2);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
typedef void (*march_fn)(const float*, float*, const AffineParams);
struct MarchCfg { int th, tw, g, la, nt; };
static const MarchCfg kMarch[] = {
    // in order of preference (the planner takes the first that fits LDS)
    {16, 32, 2, 1, 256},   // 0: two pixels per thread, 128-byte store rows, shallow ring (most workgroups per CU)
    {8, 32, 2, 1, 256},    // 1: one pixel per thread, half the footprint
    {16, 64, 2, 1, 512},   // 2: 512 threads, 256-byte store rows (measured: no faster than 0)
    {16, 32, 2, 2, 256},   // 3: two groups in flight
    {16, 32, 4, 1, 256},   // 4: four planes per barrier
    {32, 64, 2, 1, 1024},  // 5: 1024 threads
    {8, 32, 2, 1, 128},    // 6: two waves per workgroup, two pixels per thread (measured: 5-20 % slower than 0)
    {16, 32, 2, 1, 128},   // 7: two waves per workgroup, four pixels per thread (measured: 512^3 sweep mean 0.233 vs 0.234-0.244, 1024^3 on par)
};
int march_config_count() { return (int)(sizeof(kMarch) / sizeof(kMarch[0])); }
void march_config(int idx, int* th, int* tw, int* g, int* la, int* nt)
{
    *th = kMarch[idx].th; *tw = kMarch[idx].tw; *g = kMarch[idx].g; *la = kMarch[idx].la; *nt = kMarch[idx].nt;
}
int march_max_it() { return kMaxIt; }

template <int TH, int TW, int G, int LA, int NT>
static march_fn pick_march(int kind)
{
    switch (kind) {
        case 0: return affine_march_zsep<0, TH, TW, G, LA, NT>;
        case 1: return affine_march_zsep<1, TH, TW, G, LA, NT>;
        default: return affine_march_zsep<2, TH, TW, G, LA, NT>;
    }
}
static march_fn march_entry(int cfg, int kind)
{
    switch (cfg) {
        case 0: return pick_march<16, 32, 2, 1, 256>(kind);
        case 1: return pick_march<8, 32, 2, 1, 256>(kind);
        case 2: return pick_march<16, 64, 2, 1, 512>(kind);
        case 3: return pick_march<16, 32, 2, 2, 256>(kind);
        case 4: return pick_march<16, 32, 4, 1, 256>(kind);
        case 5: return pick_march<32, 64, 2, 1, 1024>(kind);
        case 6: return pick_march<8, 32, 2, 1, 128>(kind);
        default: return pick_march<16, 32, 2, 1, 128>(kind);
    }
}

typedef void (*zpair_fn)(const float*, float*, const AffineParams);
struct ZpairCfg { int th, tw, la, nt; };
static const ZpairCfg kZpair[] = {
    {16, 32, 1, 256},    // 0
    {8, 32, 1, 256},     // 1
    {16, 64, 1, 512},    // 2 (measured: slower than 0)
    {32, 64, 1, 1024},   // 3 (measured: on par with 0 at 0/90 degrees, slower at 45)
    {16, 32, 1, 512},    // 4: one pixel per thread, 8 waves per workgroup
    {32, 32, 1, 512},    // 5: two pixels per thread, square tile
    // (tried: {4, 32, 1, 64} = one wave per workgroup, no barrier at all, twelve independent waves per CU: 0.277 vs 0.245 ms at
    //  0 degrees, 0.34 vs 0.27 at 10 -- the extra halo rows each wave stages cost more than the barrier does)
};
template <int TH, int TW, int LA, int NT>
static zpair_fn pick_zpair(int kind) { return kind == 1 ? affine_march_zpair<1, TH, TW, LA, NT> : affine_march_zpair<2, TH, TW, LA, NT>; }
static zpair_fn zpair_entry(int cfg, int kind)
{
    switch (cfg) {
        case 0: return pick_zpair<16, 32, 1, 256>(kind);
        case 1: return pick_zpair<8, 32, 1, 256>(kind);
        case 2: return pick_zpair<16, 64, 1, 512>(kind);
        case 3: return pick_zpair<32, 64, 1, 1024>(kind);
        case 4: return pick_zpair<16, 32, 1, 512>(kind);
        default: return pick_zpair<32, 32, 1, 512>(kind);
    }
}
int zpair_config_count() { return (int)(sizeof(kZpair) / sizeof(kZpair[0])); }
void zpair_config(int idx, int* th, int* tw, int* la, int* nt)
{
    *th = kZpair[idx].th; *tw = kZpair[idx].tw; *la = kZpair[idx].la; *nt = kZpair[idx].nt;
}

hipError_t launch_affine_zpair(int cfg, int interp, const float* src2, float* out, const AffineParams& p,
                               int grid, int lds_bytes, hipStream_t stream)
{
    zpair_fn fn = zpair_entry(cfg, interp_kind(interp));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kZpair[cfg].nt), lds_bytes, stream, src2, out, p);
    return hipGetLastError();
}

hipError_t init_march_kernels()
{
    {
        hipError_t e = init_packed_kernels();
        if (e != hipSuccess) return e;
    }
    for (int cfg = 0; cfg < zpair_config_count(); ++cfg)
        for (int kind = 1; kind < 3; ++kind) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(zpair_entry(cfg, kind)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    for (int cfg = 0; cfg < march_config_count(); ++cfg)
        for (int kind = 0; kind < 3; ++kind) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(march_entry(cfg, kind)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

// Workgroups of this kernel that one CU keeps resident (register- and LDS-limited), from the runtime's occupancy
// calculator; cached per (kernel, LDS size) -- the planner calls this on the per-call path.
int march_blocks_per_cu(bool pair, int cfg, int interp, int lds_bytes)
{
    static std::mutex mu;
    static std::unordered_map<uint64_t, int> cache;
    const int kind = interp_kind(interp);
    const uint64_t key = ((uint64_t)(pair ? 1 : 0) << 40) | ((uint64_t)cfg << 34) | ((uint64_t)kind << 32) | (uint32_t)lds_bytes;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int n = 0;
    const void* fn = pair ? reinterpret_cast<const void*>(zpair_entry(cfg, kind)) : reinterpret_cast<const void*>(march_entry(cfg, kind));
    const int nt = pair ? kZpair[cfg].nt : kMarch[cfg].nt;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, nt, (size_t)lds_bytes) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 1;
    }
    cache.emplace(key, n);
    return n;
}

hipError_t launch_affine_march(int cfg, int interp, const float* src, float* out, const AffineParams& p,
                               int grid, int lds_bytes, hipStream_t stream)
{
    march_fn fn = march_entry(cfg, interp_kind(interp));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kMarch[cfg].nt), lds_bytes, stream, src, out, p);
    return hipGetLastError();
}

#endif  // VT_LEGACY

}  // namespace vt
