// vt_kernels_packed.hip -- general-matrix transform kernel with packed 3-D footprints (gfx950).
//
// `affine_tiled` (vt_kernels_affine.hip) stages the axis-aligned bounding box of a tile's source footprint.  For a general
// 3-D rotation of a 16^3 tile that box is 27 x 28 x 32 floats (97 KB: one workgroup per CU, 5.4x the tile's volume).
// Here only the footprint itself is staged:
//   * the footprint of a TD x TH x TW tile is the parallelepiped A.[0,T-1]^3 (+ the interpolation taps); for every
//     (z, y) row of its bounding box the workgroup computes -- once, analytically -- the x-span the parallelepiped can
//     reach in that row (valid for every sub-voxel position of a tile, so one table serves all tiles of the launch),
//     aligns it to 16 bytes and packs the spans back to back (block-wide prefix sum);
//   * every thread derives once the source offsets of the 16-byte vectors it stages per tile and keeps them in registers;
//     workgroups are persistent and walk tiles in an XCD-contiguous order, so this set-up is amortised over many tiles;
//   * per tile: `global_load_lds` of the packed footprint (border vectors from a block of zeros), one barrier, gather.
//     The gather finds a tap row's LDS position through the row table (2 extra LDS reads per voxel for trilinear).
// LDS per workgroup drops ~3x (more workgroups per CU) and so does the L2 -> LDS traffic.
// Coordinates: Q32.32 fixed-point stepping along the tile depth exactly as in affine_tiled.
#include "vt_internal.h"
#include "vt_device.h"

namespace vt {

constexpr int kPackMaxIt = 16;        // <= 4096 vectors (64 KiB) per tile footprint
constexpr int kPackRowsMax = 1024;    // (z, y) rows of the bounding box

// 16 bytes per lane, global memory -> LDS at lds_addr + 16 * lane, without a register round trip.
// Written as inline assembly: with the builtin, hipcc 7.2 merges the M0 initialisations of an unrolled sequence of
// direct-to-LDS loads and every load lands on the first destination.  The caller waits with s_waitcnt vmcnt(0).
// (Hazards inside the statement: M0 is written by a SALU instruction and read by the LDS-DMA load that follows -- one wait state,
// the s_nop; the address VGPR pair comes from VALU instructions, interlocked by the hardware.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // "m0 is reserved": exactly why it is listed as clobbered
__device__ __forceinline__ void lds_dma16(const float* g, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop

template <int KIND, int TD, int TH, int TW>
__global__ __launch_bounds__(256, (KIND == 0 ? 3 : 2)) void affine_tiled_packed(const float* __restrict__ src, float* __restrict__ out,
                                                            const float* __restrict__ zeros16, int* __restrict__ queue,
                                                            const AffineParams p, const PackGeom geo)
{
    static_assert(256 % TW == 0 && TH % (256 / TW) == 0, "tile/thread mapping");
    constexpr bool CUBIC = KIND != 0;
    constexpr int HALO = CUBIC ? 1 : 0;
    constexpr int RP = 256 / TW;
    constexpr int NJ = TH / RP;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int Lz = p.Lz, Ly = p.Ly;
    const int rows = Lz * Ly;                       // <= kPackRowsMax (host-checked)
    // LDS: [rowbase: rows ints][x0: rows ints][scratch 8 ints] | footprint buffer
    int* rowbase = reinterpret_cast<int*>(lds);
    int* rowx0 = rowbase + rows;
    int* scratch = rowx0 + rows;
    float* buf = lds + p.slot_floats;               // slot_floats = table size in floats (multiple of 4)

    // ---- row spans + block-wide exclusive prefix sum (4 rows per thread) ----
    int nv[4], x0s[4];
    int local = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = tid * 4 + r;
        int mn = 0, mx = -1;
        const bool used = (row < rows) && packed_row_span(geo, row / Ly, row % Ly, &mn, &mx);
        x0s[r] = used ? (mn & ~3) : 0;
        nv[r] = used ? (((mx - x0s[r]) >> 2) + 1) : 0;
        local += nv[r];
    }
    int incl = local;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int up = __shfl_up(incl, s);
        if (lane >= s) incl += up;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    int wave_off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) wave_off += (w < wave) ? scratch[w] : 0;
    const int nvec = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    int run = wave_off + incl - local;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = tid * 4 + r;
        if (row < rows) { rowbase[row] = run; rowx0[row] = x0s[r]; }      // rowbase holds the first vector for now
        run += nv[r];
    }
    __syncthreads();

    const int ntiles = p.nTd * p.nTh * p.nTw;
    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;
    // this thread's columns relative to the tile's first voxel, M[:, 1:3].(j, kw): formed once per launch (the tile loop adds the tile's base)
    // (one column per thread only: two columns' offsets cost the 8 x 16 x 32 tile's kernel eleven spilled registers)
    double coff[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) coff[r] = fma(p.m[4 * r + 1], (double)(tid / TW), p.m[4 * r + 2] * (double)(tid % TW));
    const int64_t ostride = (int64_t)p.oH * p.oW;
    const int kw = tid % TW;
    const int jh0 = tid / TW;
    const bool fits = (nvec * 4 <= p.Lx) && (nvec <= 256 * kPackMaxIt);  // p.Lx = footprint buffer capacity in floats
#ifdef VT_EXPERIMENTS      // make EXTRA=-DVT_EXPERIMENTS: VT_EXP_NOSTORE / VT_EXP_NOLOAD / VT_EXP_NOLDS ablations
    const bool no_stores = (p.flags & (1 << 21)) != 0, no_loads = (p.flags & (1 << 22)) != 0, no_lds = (p.flags & (1 << 26)) != 0;
#else
    constexpr bool no_stores = false, no_loads = false, no_lds = false;
#endif

    // row of every vector: each thread writes the entries of its own rows' vectors (no search afterwards); the list
    // overlays the footprint buffer, which is not in use yet
    unsigned short* vrow = reinterpret_cast<unsigned short*>(buf);
    if (fits) {
        int first = wave_off + incl - local;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            for (int i = 0; i < nv[r]; ++i) vrow[first + i] = (unsigned short)row;
            first += nv[r];
        }
    }
    __syncthreads();
    // staging descriptors: element offset relative to the box origin, and the packed (z, y, x) of the vector
    int rel[kPackMaxIt], zyx[kPackMaxIt];
    if (fits) {
#pragma unroll
        for (int it = 0; it < kPackMaxIt; ++it) {
            const int v = tid + 256 * it;
            const int row = (v < nvec) ? vrow[v] : 0;
            const int Z = row / Ly, Y = row - Z * Ly;
            const int xv = rowx0[row] + 4 * (v - rowbase[row]);
            rel[it] = (Z * p.sH + Y) * p.sP + xv;
            zyx[it] = (Z << 22) | (Y << 12) | (xv & 0xfff);
        }
    }
    __syncthreads();
    // turn the table into what the gather needs: LDS float offset of column 0 of every row
    for (int row = tid; row < rows; row += 256) rowbase[row] = 4 * rowbase[row] - rowx0[row];
    __syncthreads();
    // Trilinear: the four tap rows of a voxel are rows (z, y), (z, y+1), (z+1, y), (z+1, y+1) of the box.  The table becomes pairs
    // {byte address of column 0 of row r, of row r + Ly} (it overlays the two int tables: same size), so one 16-byte read at
    // entry (z, y) returns all four row addresses and a tap pair is one add and one ds_read2_b32 away.
    int2* const rowpair = reinterpret_cast<int2*>(lds);
    if constexpr (!CUBIC) {
        const int buf_b = (int)lds_byte_address(buf);
        int2 e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            e[r] = make_int2(0, 0);
            if (row < rows) {
                e[r].x = buf_b + 4 * rowbase[row];
                e[r].y = (row + Ly < rows) ? buf_b + 4 * rowbase[row + Ly] : e[r].x;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            if (row < rows) rowpair[row] = e[r];
        }
        __syncthreads();
    }
    typedef int pair2 __attribute__((ext_vector_type(4), aligned(8)));
    typedef float tap2 __attribute__((ext_vector_type(2), aligned(4)));
    const unsigned rowpair_b = lds_byte_address(lds);
    // one trilinear sample from the packed image (the order of the seven lerps is the order of every trilinear kernel here)
    auto sample_linear = [&](int iz, int iy, int ix, float fz, float fy, float fx) -> float {
        const unsigned ta = rowpair_b + 8u * (unsigned)(__mul24(iz, Ly) + iy);
        const pair2 t = *reinterpret_cast<const __attribute__((address_space(3))) pair2*>((size_t)ta);      // {r00, r10, r01, r11}
        const unsigned xo = 4u * (unsigned)ix;
        const tap2 a00 = *reinterpret_cast<const __attribute__((address_space(3))) tap2*>((size_t)((unsigned)t.x + xo));
        const tap2 a01 = *reinterpret_cast<const __attribute__((address_space(3))) tap2*>((size_t)((unsigned)t.z + xo));
        const tap2 a10 = *reinterpret_cast<const __attribute__((address_space(3))) tap2*>((size_t)((unsigned)t.y + xo));
        const tap2 a11 = *reinterpret_cast<const __attribute__((address_space(3))) tap2*>((size_t)((unsigned)t.w + xo));
        const float x00 = fmaf(fx, a00.y - a00.x, a00.x);
        const float x01 = fmaf(fx, a01.y - a01.x, a01.x);
        const float x10 = fmaf(fx, a10.y - a10.x, a10.x);
        const float x11 = fmaf(fx, a11.y - a11.x, a11.x);
        const float y0 = fmaf(fy, x01 - x00, x00);
        const float y1 = fmaf(fy, x11 - x10, x10);
        return fmaf(fz, y1 - y0, y0);
    };

    // ---- persistent loop over tiles ----
    // Tiles are handed out from one counter per XCD (blockIdx % 8 is the XCD of a workgroup; every XCD owns a contiguous range of
    // tile ids, whole 4 x 4 x 4 super-blocks in the blocked order), as in the lane-block kernel: the tiles in flight on an XCD are
    // always the most recent consecutive ids, a compact patch whose overlapping footprints meet in that XCD's L2, however unevenly
    // the workgroups progress (tiles outside the volume cost nothing, tiles on its rim more than the others).  With static striding
    // the patch frays: [measured, 512^3, 100 random rotations] L2 hit rate of the launch 0.35, 2.19 GB of HBM-side traffic.
    // The next id is fetched while the current tile is gathered; the last workgroup to leave zeroes the counters.
    // Round 4: the id of the next tile travels through one of two LDS words that thread 0 fills while the current tile is gathered; the
    // barrier that ends a tile (the buffer is restaged next) also publishes it -- two barriers per tile instead of four --, and the tile
    // decode divides by multiply-high with host constants (the packed kernel fetches ONE id at a time: both were per-tile costs).
    int* const ctrl = scratch + 4;                               // ctrl[0], ctrl[1] (the wave sums of the set-up are dead)
    const bool plain_order = (p.flags & (1 << 23)) != 0;         // plain (d, h, w) order (VT_TILE_ORDER=0, experiments)
    const int nids = plain_order ? ntiles : blocked_tile_count(p.nTd, p.nTh, p.nTw);
    const int nSh = (p.nTh + 3) >> 2, nSw = (p.nTw + 3) >> 2;
    const int xcd = blockIdx.x & 7, per = (((nids + 63) >> 6) + 7) / 8 * 64;      // ceil: plain order's tile count is no multiple of 64
    const int id0 = xcd * per, id_cnt = max(0, min(per, nids - id0));
    int* const counter = queue + 32 * xcd;
    int nxt = 0;
    if (tid == 0) { ctrl[0] = atomicAdd(counter, 1); nxt = atomicAdd(counter, 1); }
    __syncthreads();
    int par = 0;
    for (;;) {
        const int cur = ctrl[par];                                // published by the barrier that ended the previous tile
        if (cur >= id_cnt) break;
        par ^= 1;
        // every path through a tile ends in publish_and_sync(): thread 0 stores the next id, one barrier makes it visible (and, where the
        // tile was staged, ends the reads of the buffer)
        auto publish_and_sync = [&]() {
            if (tid == 0) { ctrl[par] = nxt; nxt = atomicAdd(counter, 1); }
            __syncthreads();
        };
        const int t = id0 + cur;
        int td_i, th_i, tw_i;
        bool tile_ok = true;
        if (plain_order) {
            tw_i = t % p.nTw;
            const int t2 = t / p.nTw;
            th_i = t2 % p.nTh;
            td_i = t2 / p.nTh;
        } else {
            // blocked_tile (vt_device.h) with its two divisions as multiply-high by host constants (vt_plan.hip: the magic numbers of the
            // super-block counts along w and h; a count of 1 has no 32-bit magic number)
            const unsigned sbi = (unsigned)t >> 6, l6 = (unsigned)t & 63u;
            const unsigned s2 = nSw == 1 ? sbi : __umulhi(sbi, p.nTw_magic), sbw = sbi - s2 * (unsigned)nSw;
            const unsigned sbd = nSh == 1 ? s2 : __umulhi(s2, p.nTh_magic), sbh = s2 - sbd * (unsigned)nSh;
            td_i = (int)(sbd * 4 + (l6 >> 4)); th_i = (int)(sbh * 4 + ((l6 >> 2) & 3)); tw_i = (int)(sbw * 4 + (l6 & 3));
            tile_ok = td_i < p.nTd && th_i < p.nTh && tw_i < p.nTw;
        }
        if (!tile_ok) { publish_and_sync(); continue; }
        const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
        const int nd = min(TD, p.oD - d0);

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
        if (!any_valid) {
            if (!keep) {
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int h = h0 + jh0 + jj * RP, w = w0 + kw;
                    if (h < p.oH && w < p.oW) {
                        float* optr = out + ((int64_t)d0 * p.oH + h) * p.oW + w;
                        for (int i = 0; i < nd; ++i) optr[i * ostride] = 0.0f;
                    }
                }
            }
        } else {
            int o[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) o[r] = (int)floor(lo[r]) - HALO;
            o[2] &= ~3;
            double b[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) b[r] = base[r] - (double)o[r];

            const unsigned buf_addr0 = lds_byte_address(buf);
            if (fits) {
                // stage the packed footprint
                const bool box_inside = o[0] >= 0 && o[1] >= 0 && o[2] >= 0 && o[0] + Lz <= p.sD && o[1] + Ly <= p.sH &&
                                        o[2] + geo.Lxbox <= p.sP;
                const int64_t origin = ((int64_t)o[0] * p.sH + o[1]) * p.sP + o[2];
                const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
                const unsigned buf_addr = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) void*)buf);
                if (box_inside) {
                    // the whole box is inside the volume (all but the rim): one descriptor based at the box origin, the vector's byte
                    // offset as the instruction's vector offset -- no 64-bit address arithmetic and no bounds test per vector
                    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<char*>(reinterpret_cast<const char*>(src + origin)), 0, 0x7fffffff, 0x00020000);
                    char* dst = reinterpret_cast<char*>(buf) + 16 * wave_first;
#pragma unroll
                    for (int it = 0; it < kPackMaxIt; ++it) {
                        if (wave_first + 256 * it < nvec) {           // wave-uniform
                            if (tid + 256 * it < nvec && !no_loads)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + 4096 * it), 16, 4 * rel[it], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < kPackMaxIt; ++it) {
                        const int v0 = wave_first + 256 * it;
                        if (v0 < nvec) {                              // wave-uniform
                            const int gz = o[0] + (zyx[it] >> 22), gy = o[1] + ((zyx[it] >> 12) & 0x3ff), gx = o[2] + (zyx[it] & 0xfff);
                            const bool inb = (unsigned)gz < (unsigned)p.sD && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sP;
                            const float* g = inb ? src + origin + rel[it] : zeros16;
                            if (tid + 256 * it < nvec && !no_loads) lds_dma16(g, buf_addr + 16u * (unsigned)v0);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the direct-to-LDS loads are invisible to hipcc's counters
                __syncthreads();                                  // drains the direct-to-LDS loads

                // Trilinear tiles wholly inside the output and the valid interval (all but the rim of the volume): no per-voxel
                // tests, 64-bit coordinate steps (one add each), stores through a buffer descriptor whose scalar offset walks the
                // planes, four voxels per loop trip so that the table reads, the tap reads and the lerps of neighbouring voxels
                // interleave.  Same arithmetic, same bits as the loop below.
                                const bool tile_fast = !CUBIC && !no_lds && all_valid && nd == TD && (h0 + TH <= p.oH) && (w0 + TW <= p.oW) &&
                                       (int64_t)TD * ostride * 4 < 0x7fffffffLL;
                if (tile_fast) {
                    if constexpr (!CUBIC) {
                        __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                            reinterpret_cast<char*>(out + ((int64_t)d0 * ostride + (int64_t)h0 * p.oW + w0)), 0, 0x7fffffff, 0x00020000);
                        const int oplane_b = (int)(ostride * 4);
                        const uint64_t inc0 = ((uint64_t)(uint32_t)p.inc_hi[0] << 32) | p.inc_lo[0];
                        const uint64_t inc1 = ((uint64_t)(uint32_t)p.inc_hi[1] << 32) | p.inc_lo[1];
                        const uint64_t inc2 = ((uint64_t)(uint32_t)p.inc_hi[2] << 32) | p.inc_lo[2];
#pragma unroll
                        for (int jj = 0; jj < NJ; ++jj) {
                            const int j = jh0 + jj * RP;
                            const Fx f0 = NJ == 1 ? to_fx(b[0] + coff[0]) : to_fx(fma(p.m[1], (double)j, fma(p.m[2], (double)kw, b[0])));
                            const Fx f1 = NJ == 1 ? to_fx(b[1] + coff[1]) : to_fx(fma(p.m[5], (double)j, fma(p.m[6], (double)kw, b[1])));
                            const Fx f2 = NJ == 1 ? to_fx(b[2] + coff[2]) : to_fx(fma(p.m[9], (double)j, fma(p.m[10], (double)kw, b[2])));
                            uint64_t c0 = ((uint64_t)(uint32_t)f0.hi << 32) | f0.lo;
                            uint64_t c1 = ((uint64_t)(uint32_t)f1.hi << 32) | f1.lo;
                            uint64_t c2 = ((uint64_t)(uint32_t)f2.hi << 32) | f2.lo;
                            const int ob = (j * p.oW + kw) * 4;
                            int soff = 0;
                            constexpr int U = 2;             // voxels in flight: 4 table reads, then 16 tap reads, then the lerps
                            static_assert(TD % U == 0, "tile depth");
                            for (int i = 0; i < TD; i += U) {
                                unsigned xo[U];
                                float fz[U], fy[U], fx[U];
                                pair2 t[U];
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                    const unsigned ta = rowpair_b + 8u * (unsigned)(__mul24((int)(c0 >> 32), Ly) + (int)(c1 >> 32));
                                    t[u] = *reinterpret_cast<const __attribute__((address_space(3))) pair2*>((size_t)ta);
                                    unsigned hx = (unsigned)(c2 >> 32);
                                    asm("" : "+v"(hx));                       // (one shift of the high word, not a 64-bit funnel shift and a mask)
                                    xo[u] = hx << 2;
                                    fz[u] = (float)(unsigned)c0 * 0x1p-32f;
                                    fy[u] = (float)(unsigned)c1 * 0x1p-32f;
                                    fx[u] = (float)(unsigned)c2 * 0x1p-32f;
                                    c0 += inc0; c1 += inc1; c2 += inc2;
                                }
                                tap2 a[U][4];
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                    a[u][0] = *reinterpret_cast<const __attribute__((address_space(3))) tap2*>((size_t)((unsigned)t[u].x + xo[u]));
                                    a[u][1] = *reinterpret_cast<const __attribute__((address_space(3))) tap2*>((size_t)((unsigned)t[u].z + xo[u]));
                                    a[u][2] = *reinterpret_cast<const __attribute__((address_space(3))) tap2*>((size_t)((unsigned)t[u].y + xo[u]));
                                    a[u][3] = *reinterpret_cast<const __attribute__((address_space(3))) tap2*>((size_t)((unsigned)t[u].w + xo[u]));
                                }
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                    // (the empty asm statements keep the four x-lerps scalar: packed, they cost six register moves)
                                    float x00 = fmaf(fx[u], a[u][0].y - a[u][0].x, a[u][0].x); asm("" : "+v"(x00));
                                    float x01 = fmaf(fx[u], a[u][1].y - a[u][1].x, a[u][1].x); asm("" : "+v"(x01));
                                    float x10 = fmaf(fx[u], a[u][2].y - a[u][2].x, a[u][2].x); asm("" : "+v"(x10));
                                    float x11 = fmaf(fx[u], a[u][3].y - a[u][3].x, a[u][3].x); asm("" : "+v"(x11));
                                    const float y0 = fmaf(fy[u], x01 - x00, x00);
                                    const float y1 = fmaf(fy[u], x11 - x10, x10);
                                    const float val = fmaf(fz[u], y1 - y0, y0);
                                    if (!no_stores || val == 123.456f)
                                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), orsrc, ob, soff, 0);
                                    soff += oplane_b;
                                }
                            }
                        }
                    }
                } else
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
                    for (int i = 0; i < nd; ++i) {
                        const int iz = c0.hi, iy = c1.hi, ix = c2.hi;
                        const float fz = fx_frac(c0), fy = fx_frac(c1), fx = fx_frac(c2);
                        float val;
                        if (no_lds) {
                            val = fz + fy + fx;
                        } else if constexpr (!CUBIC) {
                            val = sample_linear(iz, iy, ix, fz, fy, fx);
                        } else {
                            float wx[4], wy[4], wz[4];
                            cubic_weights<KIND == 2>(fx, wx);
                            cubic_weights<KIND == 2>(fy, wy);
                            cubic_weights<KIND == 2>(fz, wz);
                            // rows of the packed image start at multiples of 4 floats, so column e is 8-byte aligned
                            const int* tr = rowbase + (__mul24(iz - 1, Ly) + (iy - 1));
                            const int x1 = ix - 1, par = x1 & 1, e = x1 - par;
                            const unsigned ba = buf_addr0 + 4u * (unsigned)e;
                            int roff[16];
#pragma unroll
                            for (int c = 0; c < 4; ++c)
#pragma unroll
                                for (int bb = 0; bb < 4; ++bb) roff[4 * c + bb] = tr[c * Ly + bb];
                            val = cubic_gather_b64([&](int c, int bb) { return ba + 4u * (unsigned)roff[4 * c + bb]; }, par, wx, wy, wz);
                        }
                        const bool inside = all_valid || canonical_inside(p, d0 + i, h, w);
                        if (no_stores) { if (val == 123.456f) optr[i * ostride] = val; }
                        else if (inside) optr[i * ostride] = val;
                        else if (!keep) optr[i * ostride] = 0.0f;
                        fx_step(c0, p.inc_hi[0], p.inc_lo[0]);
                        fx_step(c1, p.inc_hi[1], p.inc_lo[1]);
                        fx_step(c2, p.inc_hi[2], p.inc_lo[2]);
                    }
                }
                // (the barrier of publish_and_sync below ends the reads of the buffer: it is restaged by the next tile)
            } else {
                // the packed footprint does not fit the buffer planned on the host: gather from global memory
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int j = jh0 + jj * RP;
                    const int h = h0 + j, w = w0 + kw;
                    if (h >= p.oH || w >= p.oW) continue;
                    for (int i = 0; i < nd; ++i) {
                        const int d = d0 + i;
                        double s[3];
#pragma unroll
                        for (int r = 0; r < 3; ++r) s[r] = canonical_coord(p, r, d, h, w);
                        const bool inside = canonical_inside(p, d, h, w);
                        float* optr = out + ((int64_t)d * p.oH + h) * p.oW + w;
                        if (inside) {
                            const double fzd = floor(s[0]), fyd = floor(s[1]), fxd = floor(s[2]);
                            *optr = direct_sample<KIND>(src, p, (int)fzd, (int)fyd, (int)fxd, (float)(s[0] - fzd), (float)(s[1] - fyd), (float)(s[2] - fxd));
                        } else if (!keep) *optr = 0.0f;
                    }
                }
            }
        }
        publish_and_sync();
    }
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(&queue[256], 1) == (int)gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) queue[32 * i] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
typedef void (*packed_fn)(const float*, float*, const float*, int*, const AffineParams, const PackGeom);
struct PackCfg { int td, th, tw; };
static const PackCfg kPack[] = {
    {8, 16, 32},
    {16, 16, 16},
    {8, 8, 32},
};
int packed_config_count() { return (int)(sizeof(kPack) / sizeof(kPack[0])); }
void packed_config(int idx, int* td, int* th, int* tw) { *td = kPack[idx].td; *th = kPack[idx].th; *tw = kPack[idx].tw; }
int packed_rows_max() { return kPackRowsMax; }
int packed_vectors_max() { return 256 * kPackMaxIt; }

template <int TD, int TH, int TW>
static packed_fn pick_packed(int kind)
{
    switch (kind) {
        case 0: return affine_tiled_packed<0, TD, TH, TW>;
        case 1: return affine_tiled_packed<1, TD, TH, TW>;
        default: return affine_tiled_packed<2, TD, TH, TW>;
    }
}
static packed_fn packed_entry(int cfg, int kind)
{
    switch (cfg) {
        case 0: return pick_packed<8, 16, 32>(kind);
        case 1: return pick_packed<16, 16, 16>(kind);
        default: return pick_packed<8, 8, 32>(kind);
    }
}

hipError_t init_packed_kernels()
{
    for (int cfg = 0; cfg < packed_config_count(); ++cfg)
        for (int kind = 0; kind < 3; ++kind) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(packed_entry(cfg, kind)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

hipError_t launch_affine_packed(int cfg, int interp, const float* src, float* out, const float* zeros16, int* queue,
                                const AffineParams& p, const PackGeom& geo, int grid, int lds_bytes, hipStream_t stream)
{
    packed_fn fn = packed_entry(cfg, interp_kind(interp));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds_bytes, stream, src, out, zeros16, queue, p, geo);
    return hipGetLastError();
}

}  // namespace vt
