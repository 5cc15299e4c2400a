// vt_kernels_span.hip -- trilinear transform under a general matrix (true 3-D rotations: the reference's own benchmark protocol,
// tests/benchmark.py:52-54): packed row spans of the tile's source footprint in LDS, kernel id 6 (gfx950).  Round 5.
//
// Same algorithm as round 1's `affine_tiled_packed` (vt_kernels_packed.hip, which keeps the cubic instantiations): the footprint of a
// TD x TH x TW output tile is the parallelepiped A.[0,T-1]^3; per (z, y) row of its bounding box only the x-span the parallelepiped can
// reach is staged (one span table per launch: packed_row_span, vt_internal.h, valid for every sub-voxel position of a tile), rows packed
// back to back, and the gather finds a tap row through a table.  tools/footprint_survey.py says why the table stays: over the reference's
// 100 random rotations the packed spans hold 1.6-2.0 floats per output voxel, a box with row starts that are LINEAR in (z, y) ("sheared
// box", the alternative that needs no table) 3.3-5.1, the bounding box 4.1-7.3.
//
// What round 5 changes is everything around the table.  The counters of the old kernel (profiles/r04_general512_linear_summary.json) say
// 74 wave-instructions per 64 voxels, of which the voxel loop is 41: the other 33 are the per-tile prologue -- tile geometry in float64
// done per lane, ~100 v_readlane of spilled scalars, a waterfall loop around every LDS-DMA load (the buffer descriptor lived in vector
// registers: its base came from a float64 -> int conversion) -- at 167 VGPRs = 3 waves per SIMD.  Here:
//   * the tile's origin becomes scalar right after the conversion (v_readfirstlane through the builtin), so box tests, descriptors and
//     LDS-DMA addresses are scalar arithmetic; one `buffer_load_dwordx4 ... lds` per staged vector and nothing else;
//   * the launch's float64 constants are re-read per tile from the kernel-argument segment (scalar cache) instead of being pinned in
//     100 SGPRs and spilled; a thread's sub-tile coordinate offset is a Q32.32 constant formed once per launch, a tile adds its base;
//   * table entries are 16 bytes {row (z,y), (z,y+1), (z+1,y), (z+1,y+1)}: one aligned ds_read_b128 instead of a ds_read2_b64;
//   * the seven lerps run y -> z -> x: the pairs (x, x+1) that ds_read2_b32 returns are the operands of packed FMAs (v_pk_fma_f32:
//     two lerps per instruction; gfx950 issues a wave64 vector instruction over 4 cycles and a packed one does twice the work in them);
//   * 64-bit coordinate steps are one v_lshl_add_u64 each; four voxels in flight per thread;
//   * rim tiles (box not wholly inside the volume, or voxels beyond the skirt / the output) go through the SAME gather with a store mask
//     formed ahead of it by the canonical float64 test (vt_device.h), so classification never depends on the kernel;
//   * <= 128 VGPRs: four workgroups per CU where the footprint fits 40 KiB.
// Arithmetic: coordinates as in every tiled kernel here (float64 tile base, Q32.32 steps); the lerp order differs from the other trilinear
// kernels (x last instead of first), i.e. by float32 rounding of a convex combination -- tests hold it to the family's 1e-6.
#include "vt_internal.h"
#include "vt_device.h"

namespace vt {

constexpr int kSpanMaxIt = 16;        // staging vectors per thread: footprints up to 4096 vectors (64 KiB)
constexpr int kSpanRowsMax = 1024;    // (z, y) rows of the bounding box

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float tap2 __attribute__((ext_vector_type(2), aligned(4)));

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // "m0 is reserved": exactly why it is listed as clobbered
// One LDS-DMA vector per lane: 16 bytes from descriptor `rs` at byte offset `voff` to LDS address m0 + 16 * lane.
// (M0 is written by a SALU instruction and read by the load: one wait state, the s_nop.  The offset VGPR comes from VALU
// instructions long before, interlocked by the hardware anyway.  The caller waits with s_waitcnt vmcnt(0).)
__device__ __forceinline__ void span_dma16(int voff, v4i rs, unsigned m0v)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rs), "s"(m0v) : "memory", "m0");
}
__device__ __forceinline__ void span_dma16_global(const float* g, unsigned m0v)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(m0v) : "memory", "m0");
}
#pragma clang diagnostic pop

// Fetch the next tile id from a queue counter WITHOUT waiting for it.  Written as a plain atomicAdd() by one lane, hipcc's atomic optimiser
// (the address is uniform) turns it into a wave-aggregated atomic followed at once by `s_waitcnt vmcnt(0)` + v_readfirstlane -- it needs
// the old value to hand every lane its share -- and that wait also covers the sixteen output stores the wave has just issued:
// [measured with s_memtime stamps, 512^3] 4 100 cycles per tile between the last store and the end of the publishing barrier, 16 % of the
// kernel, every wave of the workgroup waiting at the barrier for lane 0's stores to be acknowledged.  With an address the compiler cannot
// prove uniform the atomic stays a plain returning atomic, and the wait moves to the first use of the value, a tile later, where the
// compiler counts the stores issued since (`vmcnt(N)`: the LDS-DMA loads of the staging are older than those stores, so the count holds).
__device__ __forceinline__ int queue_fetch_async(int* counter)
{
    typedef __attribute__((address_space(1))) int* global_int_ptr;            // (stays a global_ instruction: a flat_ one counts in lgkmcnt too)
    global_int_ptr g = (global_int_ptr)counter;
    asm volatile("" : "+v"(g));
    return __hip_atomic_fetch_add(g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a value the compiler cannot prove uniform (so that it keeps the v_readfirstlane and the result lives in a scalar register)
__device__ __forceinline__ int to_scalar(int v)
{
    asm volatile("" : "+v"(v));
    return __builtin_amdgcn_readfirstlane(v);
}

// Q32.32 with floor semantics (two's complement for negative x), rounded to the NEAREST multiple of 2^-32.  A thread's coordinate is the sum of
// three such values (tile base, the thread's column offset, i steps): with truncation a column offset of -2e-16 (what cos(90 degrees) in
// float32 leaves behind) becomes -2.3e-10, and a coordinate that is an exact integer -- quarter turns, integer shifts -- lands one row
// below the box.  Rounded to nearest, every term of an integer-valued coordinate is exact.  What is left is the rounding itself, at most
// (2 + TD) x 2^-33 < 2.1e-9 per coordinate: the box origin is taken 4e-9 below the tile's lowest coordinate and the planner sizes the box
// for an extent 1e-8 larger (vt_plan.hip: pick_packed_tile), so a coordinate within that distance of a box face still indexes inside the box.
__device__ __forceinline__ uint64_t fx64(double x)
{
    const double fl = floor(x);
    double fr = rint((x - fl) * 4294967296.0);
    int hi = (int)fl;
    if (fr >= 4294967296.0) { fr = 0.0; hi += 1; }
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)fr;
}
constexpr double kSpanOriginMargin = 4.0e-9;

// One column of TD voxels along the tile's depth: Q32.32 box coordinates (c0, c1, c2) of its first voxel, stepped by (inc0, inc1, inc2).
// Four voxels in flight: their table reads, then their sixteen tap-pair reads, then the lerps and stores.  MASKED (rim tiles): bit i of
// `am` says voxel i stores its value, of `zm` a zero (the canonical test ran ahead of the gather); the reads happen regardless -- an LDS
// address beyond the allocation returns 0, and nothing is stored from it.
// `bdelta`: byte distance of the buffer to read from the one the table's addresses refer to (the pipelined kernel's second buffer).
template <int TD, bool MASKED, int UNROLL = 1>
__device__ __forceinline__ void span_gather_column(uint64_t c0, uint64_t c1, uint64_t c2, uint64_t inc0, uint64_t inc1, uint64_t inc2,
                                                   unsigned tbl_b, int Ly, __amdgpu_buffer_rsrc_t orsrc, int obj, int oplane_b, unsigned am, unsigned zm,
                                                   bool no_stores = false, bool no_lds = false, unsigned bdelta = 0u)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    int soff = 0;                                    // scalar: byte offset of the group's first output plane
#pragma unroll UNROLL
    for (int i0 = 0; i0 < TD; i0 += 4) {             // (a real loop unless the tile is shallow: unrolled, the compiler forms all TD coordinates up front and spills)
        v4i t[4];
        unsigned xo[4];
        float fz[4], fy[4], fx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int iz = (int)(c0 >> 32), iy = (int)(c1 >> 32);
            const unsigned ix = (unsigned)(c2 >> 32);
            const unsigned ta = tbl_b + 16u * (unsigned)(__mul24(iz, Ly) + iy);
            if (no_lds) t[u] = (v4i)((int)ta);
            else t[u] = *reinterpret_cast<const __attribute__((address_space(3))) v4i*>((size_t)ta);
            xo[u] = 4u * ix + bdelta;
            fz[u] = (float)(uint32_t)c0 * 0x1p-32f;
            fy[u] = (float)(uint32_t)c1 * 0x1p-32f;
            fx[u] = (float)(uint32_t)c2 * 0x1p-32f;
            c0 += inc0; c1 += inc1; c2 += inc2;
            asm("" : "+v"(c0), "+v"(c1), "+v"(c2));           // (one chain of three 64-bit adds per voxel: left alone, loop strength reduction keeps twelve induction variables)
        }
        tap2 a[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (no_lds) { a[u][k].x = fz[u] + (float)k; a[u][k].y = fy[u]; }
                else a[u][k] = *reinterpret_cast<const __attribute__((address_space(3))) tap2*>((size_t)((unsigned)t[u][k] + xo[u]));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const v2f a00 = {a[u][0].x, a[u][0].y}, a01 = {a[u][1].x, a[u][1].y};
            const v2f a10 = {a[u][2].x, a[u][2].y}, a11 = {a[u][3].x, a[u][3].y};
            const v2f y0 = __builtin_elementwise_fma((v2f)(fy[u]), a01 - a00, a00);
            const v2f y1 = __builtin_elementwise_fma((v2f)(fy[u]), a11 - a10, a10);
            const v2f zz = __builtin_elementwise_fma((v2f)(fz[u]), y1 - y0, y0);
            const float val = fmaf(fx[u], zz.y - zz.x, zz.x);
            if constexpr (!MASKED) {
                if (!no_stores || val == 123.456f)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), orsrc, obj, soff + u * oplane_b, 0);
            } else {
                const bool sv = (am >> u) & 1u, sz = (zm >> u) & 1u;
                if (sv || sz) __builtin_amdgcn_raw_buffer_store_b32(sv ? __builtin_bit_cast(unsigned, val) : 0u, orsrc, obj, soff + u * oplane_b, 0);
            }
        }
        soff += 4 * oplane_b;
        if constexpr (MASKED) { am >>= 4; zm >>= 4; }
    }
}

template <int TD, int TH, int TW>
__global__ __launch_bounds__(256, 4) void affine_span(const float* __restrict__ src, float* __restrict__ out, const float* __restrict__ zeros16,
                                                      int* __restrict__ queue, const AffineParams p, const PackGeom geo)
{
    static_assert(256 % TW == 0 && TH % (256 / TW) == 0 && TD % 4 == 0, "tile/thread mapping");
    constexpr int RP = 256 / TW;                     // tile rows covered by the 256 threads
    constexpr int NJ = TH / RP;                      // columns (h, w) per thread
    extern __shared__ __attribute__((aligned(16))) float lds[];

#ifdef VT_EXPERIMENTS      // make EXTRA=-DVT_EXPERIMENTS: VT_EXP_NOSTORE / VT_EXP_NOLOAD / VT_EXP_NOLDS ablations
    const bool no_stores = (p.flags & (1 << 21)) != 0, no_loads = (p.flags & (1 << 22)) != 0, no_lds = (p.flags & (1 << 26)) != 0;
    const bool no_loop = (p.flags & (1 << 27)) != 0, static_ids = (p.flags & (1 << 28)) != 0;
    const bool no_tiles = (p.flags & (1 << 30)) != 0;              // VT_EXP_NOTILES: the workgroup's set-up and nothing else
    // VT_EXP_STAMPS: s_memtime at the phase boundaries of the tile loop, summed per workgroup and written over the first floats of the
    // output at the end (a diagnostic build: the output is garbage there anyway); tools/span_probe.py prints the shares
    const bool stamps = (p.flags & (1 << 29)) != 0;
    uint64_t st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int st_rim = 0, st_cnt[3] = {0, 0, 0};
    uint64_t st_prev = __builtin_amdgcn_s_memtime();
    const uint64_t st_begin = st_prev;
#define VT_STAMP(k) if (stamps) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); st_acc[(k) + st_rim] += now_ - st_prev; st_prev = now_; }
#else
#define VT_STAMP(k)
    constexpr bool no_stores = false, no_loads = false, no_lds = false, no_loop = false, static_ids = false, no_tiles = false;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int Lz = p.Lz, Ly = p.Ly;
    const int rows = Lz * Ly;                        // <= kSpanRowsMax (host-checked)
    // LDS: [table: rows x 16 bytes][4 control words] | footprint buffer (p.slot_floats floats in)
    int* const tbl = reinterpret_cast<int*>(lds);
    int* const ctrl = tbl + 4 * rows;
    float* const buf = lds + p.slot_floats;
    const unsigned tbl_b = lds_byte_address(lds), buf_b = lds_byte_address(buf);

    // ---- once per workgroup: row spans, their prefix sum (4 rows per thread), the staging descriptors, the gather table ----
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
    const int incl = wave_scan_add(local);
    if (lane == 63) ctrl[wave] = incl;
    __syncthreads();
    int wave_off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) wave_off += (w < wave) ? ctrl[w] : 0;
    const int nvec = ctrl[0] + ctrl[1] + ctrl[2] + ctrl[3];
    const int nvec_pad = (nvec + 63) & ~63;          // whole waves stage: the lanes behind the last vector re-read vector 0's source
    {
        int run = wave_off + incl - local;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            if (row < rows) { tbl[4 * row] = run; tbl[4 * row + 1] = x0s[r]; }     // {first vector, first column} for now
            run += nv[r];
        }
    }
    const bool fits = (nvec_pad * 4 <= p.Lx) && (nvec_pad <= 256 * kSpanMaxIt);     // p.Lx = footprint buffer capacity in floats
    // row of every vector: each thread writes the entries of its own rows' vectors; the list overlays the footprint buffer
    unsigned short* const vrow = reinterpret_cast<unsigned short*>(buf);
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
    // staging descriptors: byte offset of the vector relative to the box origin, -1 for the lanes behind the last vector.  (Rim tiles
    // need the vector's (z, y, x) inside the box for their bounds tests: they divide the offset by the plane and row sizes again
    // instead of keeping a second descriptor per vector in registers all the time -- the box is no larger than the volume in y and x,
    // host-checked, so the division is the inverse.)
    int rel[kSpanMaxIt];
#pragma unroll
    for (int it = 0; it < kSpanMaxIt; ++it) {
        const int v = tid + 256 * it;
        const bool real = fits && v < nvec;
        const int row = real ? vrow[v] : 0;
        const int Z = row / Ly, Y = row - Z * Ly;
        const int xv = real ? tbl[4 * row + 1] + 4 * (v - tbl[4 * row]) : 0;
        rel[it] = real ? ((Z * p.sH + Y) * p.sP + xv) * 4 : -1;         // < 2^31 (host-checked: Lz planes of the source)
    }
    __syncthreads();
    // the table the gather reads: entry (z, y) = LDS byte addresses of column 0 of rows (z, y), (z, y+1), (z+1, y), (z+1, y+1)
    {
        v4i e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            e[r] = (v4i)(0);
            if (row < rows) {
                auto addr = [&](int rr) { return (int)buf_b + 4 * (4 * tbl[4 * rr] - tbl[4 * rr + 1]); };
                const int y = row % Ly;
                const int r01 = (y + 1 < Ly) ? row + 1 : row, r10 = (row + Ly < rows) ? row + Ly : row;
                const int r11 = (y + 1 < Ly && row + Ly < rows) ? row + Ly + 1 : r10;
                e[r] = (v4i){addr(row), addr(r01), addr(r10), addr(r11)};
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            if (row < rows) *reinterpret_cast<v4i*>(tbl + 4 * row) = e[r];
        }
    }

    // this thread's columns: Q32.32 offset of the column's first voxel from the tile's, M[:, 1:3].(j, kw), and the byte offset in the output
    const int kw = tid % TW, jh0 = tid / TW;
    uint64_t coff[NJ][3];
    int ob[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = jh0 + jj * RP;
#pragma unroll
        for (int r = 0; r < 3; ++r) coff[jj][r] = fx64(fma(p.m[4 * r + 1], (double)j, p.m[4 * r + 2] * (double)kw));
        ob[jj] = (j * p.oW + kw) * 4;
    }
    const uint64_t inc0 = ((uint64_t)(uint32_t)p.inc_hi[0] << 32) | p.inc_lo[0];
    const uint64_t inc1 = ((uint64_t)(uint32_t)p.inc_hi[1] << 32) | p.inc_lo[1];
    const uint64_t inc2 = ((uint64_t)(uint32_t)p.inc_hi[2] << 32) | p.inc_lo[2];
    const int64_t ostride = (int64_t)p.oH * p.oW;
    const int oplane_b = (int)(ostride * 4);         // TD planes of the output stay below 2^31 bytes (host-checked)
    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;
    VT_STAMP(0)                                       // 0: set-up of the workgroup

    // ---- persistent loop over tiles: one counter per XCD, one id per fetch, blocked 4 x 4 x 4 order (see vt_kernels_packed.hip) ----
    const bool plain_order = (p.flags & (1 << 23)) != 0;
    const int ntiles = p.nTd * p.nTh * p.nTw;
    const int nids = plain_order ? ntiles : blocked_tile_count(p.nTd, p.nTh, p.nTw);
    const int nSh = (p.nTh + 3) >> 2, nSw = (p.nTw + 3) >> 2;
    const int xcd = blockIdx.x & 7, per = (((nids + 63) >> 6) + 7) / 8 * 64;
    const int id0 = xcd * per, id_cnt = no_tiles ? 0 : max(0, min(per, nids - id0));
    int* const counter = queue + 32 * xcd;
    int nxt = 0;
    __syncthreads();                                  // the table is complete, the wave sums in ctrl[] are dead
    const int nper = (int)gridDim.x >> 3, mine = (int)blockIdx.x >> 3;       // (experiment: static striding inside the XCD's id range)
    if (tid == 0) {
        if (static_ids) { ctrl[0] = mine; nxt = mine + nper; }
        else { ctrl[0] = atomicAdd(counter, 1); nxt = queue_fetch_async(counter); }
    }
    __syncthreads();
    int par = 0;
    if (!fits) {
        // The packed footprint does not fit the buffer planned on the host (never seen: the plan carries a margin over the host's own
        // count): this workgroup gathers its tiles from global memory, ids from the same queue.
        for (;;) {
            const int cur = ctrl[par];
            if (cur >= id_cnt) break;
            par ^= 1;
            int td_i, th_i, tw_i;
            const bool tile_ok = plain_order ? (td_i = (id0 + cur) / (p.nTh * p.nTw), th_i = ((id0 + cur) / p.nTw) % p.nTh, tw_i = (id0 + cur) % p.nTw, true)
                                             : blocked_tile(id0 + cur, p.nTd, p.nTh, p.nTw, td_i, th_i, tw_i);
            if (tile_ok) {
                const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
#pragma unroll 1
                for (int jj = 0; jj < NJ; ++jj) {
                    const int h = h0 + jh0 + jj * RP, w = w0 + kw;
                    if (h >= p.oH || w >= p.oW) continue;
                    const int nd = min(TD, p.oD - d0);
#pragma unroll 1
                    for (int i = 0; i < nd; ++i) {
                        const int d = d0 + i;
                        float* optr = out + ((int64_t)d * p.oH + h) * p.oW + w;
                        if (canonical_inside(p, d, h, w)) {
                            double sc[3];
#pragma unroll
                            for (int r = 0; r < 3; ++r) sc[r] = canonical_coord(p, r, d, h, w);
                            const double fzd = floor(sc[0]), fyd = floor(sc[1]), fxd = floor(sc[2]);
                            *optr = direct_sample<0>(src, p, (int)fzd, (int)fyd, (int)fxd, (float)(sc[0] - fzd), (float)(sc[1] - fyd), (float)(sc[2] - fxd));
                        } else if (!keep) *optr = 0.0f;
                    }
                }
            }
            if (tid == 0) { ctrl[par] = nxt; nxt = atomicAdd(counter, 1); }
            __syncthreads();
        }
    } else
    for (;;) {
        const int cur = __builtin_amdgcn_readfirstlane(ctrl[par]);      // published by the barrier that ended the previous tile
        if (cur >= id_cnt) break;
        par ^= 1;
        // The launch constants are read afresh for every tile from the kernel-argument segment (scalar loads that hit the scalar cache;
        // the four pointer arguments are 32 bytes, AffineParams and PackGeom follow): hoisted out of the tile loop they are ~100 scalar
        // registers that stay live across the voxel loop, and the allocator's spills come back through v_readlane at every use.
        typedef const __attribute__((address_space(4))) char* KArg;
        KArg ka = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const auto* const kp = (const __attribute__((address_space(4))) AffineParams*)(ka + 32);
        const auto* const kg = (const __attribute__((address_space(4))) PackGeom*)(ka + 32 + sizeof(AffineParams));
        // what this id turns out to be: 0 = no tile (padding id of the blocked order), 1 = a tile wholly outside the valid interval
        // (zero-fill), 2 = a tile whose footprint has been staged
        int mode = 0;
        bool fast = false, whole = false;
        int d0 = 0, h0 = 0, w0 = 0;
        uint64_t bfx[3] = {0, 0, 0};
        unsigned inm[NJ], zm[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) { inm[jj] = 0xffffffffu; zm[jj] = 0u; }
        __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out), 0, 0, 0x00020000);
        do {
            const int t = id0 + cur;
            int td_i, th_i, tw_i;
            bool tile_ok = true;
            if (plain_order) {
                tw_i = t % kp->nTw;
                const int t2 = t / kp->nTw;
                th_i = t2 % kp->nTh;
                td_i = t2 / kp->nTh;
            } else {
                const unsigned sbi = (unsigned)t >> 6, l6 = (unsigned)t & 63u;
                const unsigned s2 = nSw == 1 ? sbi : __umulhi(sbi, kp->nTw_magic), sbw = sbi - s2 * (unsigned)nSw;
                const unsigned sbd = nSh == 1 ? s2 : __umulhi(s2, kp->nTh_magic), sbh = s2 - sbd * (unsigned)nSh;
                td_i = (int)(sbd * 4 + (l6 >> 4)); th_i = (int)(sbh * 4 + ((l6 >> 2) & 3)); tw_i = (int)(sbw * 4 + (l6 & 3));
                tile_ok = td_i < kp->nTd && th_i < kp->nTh && tw_i < kp->nTw;
            }
            if (!tile_ok) break;
            d0 = td_i * TD; h0 = th_i * TH; w0 = tw_i * TW;
            VT_STAMP(1)                               // 1: id, decode

            // ---- tile geometry: float64 on wave-uniform values ----
            double base[3], lo[3];
            bool any_valid = true, all_valid = true;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                base[r] = fma(kp->m[4 * r], (double)d0, fma(kp->m[4 * r + 1], (double)h0, fma(kp->m[4 * r + 2], (double)w0, kp->m[4 * r + 3])));
                lo[r] = base[r] + kp->neg[r];
                const double hi = base[r] + kp->pos[r];
                any_valid = any_valid && (hi >= kp->vlo[r] - kTileMargin) && (lo[r] < kp->vhi[r] + kTileMargin);
                all_valid = all_valid && (lo[r] >= kp->vlo[r] + kTileMargin) && (hi < kp->vhi[r] - kTileMargin);
            }
            whole = (d0 + TD <= kp->oD) && (h0 + TH <= kp->oH) && (w0 + TW <= kp->oW);
            const int64_t otile = (int64_t)d0 * ostride + (int64_t)h0 * kp->oW + w0;
            orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out + otile), 0, 0x7fffffff, 0x00020000);
            mode = 1;
            if (!any_valid) break;                    // the whole tile maps outside the valid interval
            mode = 2;

            // box origin, scalar: floor(lowest source coordinate of the tile), x aligned down to a 16-byte vector
            const int o0 = to_scalar((int)floor(lo[0] - kSpanOriginMargin)), o1 = to_scalar((int)floor(lo[1] - kSpanOriginMargin)),
                      o2 = to_scalar((int)floor(lo[2] - kSpanOriginMargin)) & ~3;
            const bool box_inside = o0 >= 0 && o1 >= 0 && o2 >= 0 && o0 + Lz <= kp->sD && o1 + Ly <= kp->sH && o2 + kg->Lxbox <= kp->sP;
            const int64_t origin = ((int64_t)o0 * kp->sH + o1) * kp->sP + o2;
            VT_STAMP(2)                               // 2: geometry up to the scalar box origin
#ifdef VT_EXPERIMENTS
            st_rim = (all_valid && whole && box_inside) ? 0 : 8;          // phases 3..6 of rim tiles are counted apart (indices 11..14)
            st_cnt[st_rim ? 1 : 0] += 1;
#endif

            // ---- stage the packed footprint: vector v of the list lands at buf + 16 v ----
            const unsigned m0v = buf_b + 16u * (unsigned)wave_first;
            if (box_inside) {
                // the box is inside the volume: one descriptor based at its origin, the vector's byte offset as the vector offset
                const uint64_t sb = (uint64_t)(size_t)(src + origin);
                const v4i rs = {(int)(uint32_t)sb, (int)((uint32_t)(sb >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
#pragma unroll
                for (int it = 0; it < kSpanMaxIt; ++it) {
                    if (wave_first + 256 * it < nvec_pad && !no_loads)             // wave-uniform
                        span_dma16(max(rel[it], 0), rs, m0v + 4096u * it);
                }
            } else {
                // rim: per-vector bounds tests, vectors outside the volume come from a block of zeros
                const int row_f = kp->sP, plane_f = kp->sH * row_f;               // floats per source row / plane
                const float inv_row = 1.0f / (float)row_f, inv_plane = 1.0f / (float)plane_f;
#pragma unroll 1
                for (int it = 0; it < kSpanMaxIt; ++it) {
                    if (wave_first + 256 * it >= nvec_pad) break;     // wave-uniform
                    int r4 = rel[it];
                    asm volatile("" : "+v"(r4));
                    const int u = r4 >> 2;                            // float offset inside the box, (Z * sH + Y) * sP + xv
                    int Z = (int)((float)u * inv_plane);              // estimate within one of the quotient, then corrected
                    int rem = u - Z * plane_f;
                    if (rem < 0) { --Z; rem += plane_f; }
                    if (rem >= plane_f) { ++Z; rem -= plane_f; }
                    int Y = (int)((float)rem * inv_row);
                    int xv = rem - Y * row_f;
                    if (xv < 0) { --Y; xv += row_f; }
                    if (xv >= row_f) { ++Y; xv -= row_f; }
                    const int gz = o0 + Z, gy = o1 + Y, gx = o2 + xv;
                    const bool inb = r4 >= 0 && (unsigned)gz < (unsigned)kp->sD && (unsigned)gy < (unsigned)kp->sH && (unsigned)gx < (unsigned)kp->sP;
                    const float* g = inb ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(src + origin) + r4) : zeros16;
                    span_dma16_global(g, m0v + 4096u * it);
                }
            }
            VT_STAMP(3)                               // 3: staging issued

            // this tile's sub-voxel base in Q32.32 (while the loads are in flight): box coordinate of the tile's first voxel, >= 0
            bfx[0] = fx64(base[0] - (double)o0);
            bfx[1] = fx64(base[1] - (double)o1);
            bfx[2] = fx64(base[2] - (double)o2);

            // store masks of rim tiles, formed ahead of the gather: bit i of inm = voxel i of the column stores its value, of zm = a zero
            fast = all_valid && whole;
            if (!fast) {
                // canonical_inside (vt_device.h) on the constants as read from the argument segment: the same fma chain, column order p.ord
                auto inside_canonical = [&](int d, int h, int w) {
                    const int c0 = kp->ord[0], c1 = kp->ord[1], c2 = kp->ord[2];
                    const double x0 = (double)(c0 == 0 ? d : (c0 == 1 ? h : w));
                    const double x1 = (double)(c1 == 0 ? d : (c1 == 1 ? h : w));
                    const double x2 = (double)(c2 == 0 ? d : (c2 == 1 ? h : w));
                    bool in = true;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const double sc = fma(kp->m[4 * r + c0], x0, fma(kp->m[4 * r + c1], x1, fma(kp->m[4 * r + c2], x2, kp->m[4 * r + 3])));
                        in = in && (sc >= kp->vlo[r]) && (sc < kp->vhi[r]);
                    }
                    return in;
                };
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int h = h0 + jh0 + jj * RP, w = w0 + kw;
                    unsigned a = 0u, z = 0u;
                    if (h < kp->oH && w < kp->oW) {
                        const int nd = min(TD, kp->oD - d0);
#pragma unroll 1
                        for (int i = 0; i < nd; ++i) {
                            const bool inside = all_valid || inside_canonical(d0 + i, h, w);
                            a |= inside ? (1u << i) : 0u;
                            z |= (!inside && !keep) ? (1u << i) : 0u;
                        }
                    }
                    inm[jj] = a; zm[jj] = z;
                }
            }
            VT_STAMP(4)                               // 4: sub-voxel base, store masks of rim tiles
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the direct-to-LDS loads are invisible to hipcc's counters
        } while (false);

        // ONE place where thread 0 hands the next id on (fetched a tile ago: it has arrived -- a staged tile has just waited for every
        // vector-memory operation, the other paths have issued none since) and fetches the one after, AHEAD of this tile's stores: the
        // compiler waits for an atomic's value where it is used, with `s_waitcnt vmcnt(0)` across the loop's back edge, and behind the
        // gather that wait would cover the sixteen stores a wave has just issued -- [measured with s_memtime stamps, 512^3] 4 100 cycles
        // per tile with every wave of the workgroup waiting at the barrier for wave 0's stores to be acknowledged, 16 % of the kernel.
        if (tid == 0) {
            ctrl[par] = nxt;
            if (static_ids) nxt += nper; else nxt = queue_fetch_async(counter);
        }
        if (mode == 2) {
            __syncthreads();                          // the staged bytes of every wave have landed
            VT_STAMP(5)                               // 5: wait for the staged bytes, barrier
            // ---- gather: TD voxels per column, four in flight; one loop without store masks (all but the rim), one with ----
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const uint64_t c0 = bfx[0] + coff[jj][0], c1 = bfx[1] + coff[jj][1], c2 = bfx[2] + coff[jj][2];
                if (no_loop) { if (c0 == 12345u) out[c1] = 0.f; }
                else if (fast) span_gather_column<TD, false>(c0, c1, c2, inc0, inc1, inc2, tbl_b, Ly, orsrc, ob[jj], oplane_b, 0u, 0u, no_stores, no_lds);
                else span_gather_column<TD, true>(c0, c1, c2, inc0, inc1, inc2, tbl_b, Ly, orsrc, ob[jj], oplane_b, inm[jj], zm[jj]);
            }
            VT_STAMP(6)                               // 6: gather
        } else if (mode == 1) {
            // the whole tile maps outside the valid interval: zero-fill (or leave untouched)
            if (!keep && !no_stores) {
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int h = h0 + jh0 + jj * RP, w = w0 + kw;
                    if (whole || (h < kp->oH && w < kp->oW)) {
                        const int nd = min(TD, kp->oD - d0);
                        for (int i = 0; i < nd; ++i) __builtin_amdgcn_raw_buffer_store_b32(0u, orsrc, ob[jj], i * oplane_b, 0);
                    }
                }
            }
            VT_STAMP(7)                               // 7: tiles outside the volume, whole
#ifdef VT_EXPERIMENTS
            st_cnt[2] += 1;
#endif
        }
#ifdef VT_EXPERIMENTS
        st_rim = 0;
#endif
        __syncthreads();                              // the next id is visible; the reads of the buffer are over
        VT_STAMP(1)                                   // (the publishing barrier counts with the next tile's id)
    }
#ifdef VT_EXPERIMENTS
    if (stamps && tid == 0) {
        float* dbg = out + 16 * (size_t)blockIdx.x;
        for (int k = 0; k < 8; ++k) dbg[k] = (float)st_acc[k];
        dbg[8] = (float)(__builtin_amdgcn_s_memtime() - st_begin);
        for (int k = 3; k < 7; ++k) dbg[6 + k] = (float)st_acc[8 + k];          // 9..12: phases 3..6 of rim tiles
        dbg[13] = (float)st_cnt[0]; dbg[14] = (float)st_cnt[1]; dbg[15] = (float)st_cnt[2];
    }
#endif
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(&queue[256], 1) == (int)gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) queue[32 * i] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same tiles, software-pipelined by WAVE SPECIALISATION: 320 threads -- four consumer waves that gather (and do nothing else) and one
// producer wave that fetches tile ids, works out tile geometry, stages the NEXT tile's footprint into a second LDS buffer and hands the
// tile's parameters over through LDS.  One barrier per tile.
//
// Why: [measured, 512^3, ablation build of `affine_span`] with no loads, no stores, no LDS reads and no voxel loop at all 0.21 ms of the
// 0.42 remain -- the chain id -> geometry -> staging issue -> HBM latency -> barrier -> gather -> barrier of every tile, which three or
// four workgroups per CU overlap only in part (halving the instruction count did not move the launch time).  And every attempt to keep
// loads, stores and the queue atomic of ONE wave in flight across each other ran into `s_waitcnt vmcnt`: the counter is in order and the
// compiler's own waits are conservative across loop edges (vmcnt(0) in front of a use of the atomic's value = a wait for the sixteen
// stores just issued).  Here no wave mixes the kinds: consumers issue stores and never wait for vector memory; the producer issues
// loads (and the atomic) and its vmcnt(0) is exactly "my staging has landed".
//   barrier k:   producer has seen S(k) land and has published P(k);  consumers have finished C(k-1)
//   then:        producer fetches id(k+2), computes G(k+1), issues S(k+1) into buffer (k+1)&1, publishes P(k+1), waits vmcnt(0)
//                consumers read P(k), gather C(k) from buffer k&1
// (S staging loads, G geometry, P parameters, C gather.)
// ---------------------------------------------------------------------------------------------------
constexpr int kSpanWsMaxIt = 40;      // staging vectors per PRODUCER lane: footprints up to 2560 vectors (40 KiB) per buffer (8 x 320 threads write the list)

template <int TD, int TH, int TW>
__global__ __launch_bounds__(320) void affine_span_ws(const float* __restrict__ src, float* __restrict__ out, const float* __restrict__ zeros16,
                                                      int* __restrict__ queue, const AffineParams p, const PackGeom geo)
{
    static_assert(TH * TW == 256 && TD % 4 == 0, "one column per consumer thread");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool producer = __builtin_amdgcn_readfirstlane(wave) == 4;
    const int Lz = p.Lz, Ly = p.Ly;
    const int rows = Lz * Ly;
    // LDS: [table: rows x 16 bytes][parameter slots: 4 x 20 dwords][4 scratch words] | buffer 0 | buffer 1 (p.Lx floats each) | descriptor list
    int* const tbl = reinterpret_cast<int*>(lds);
    int* const slot = tbl + 4 * rows;
    int* const scratch = slot + 80;
    float* const buf = lds + p.slot_floats;
    const unsigned tbl_b = lds_byte_address(lds), buf_b = lds_byte_address(buf);
    const unsigned bdelta = 4u * (unsigned)p.Lx;

    // ---- once per workgroup: row spans (consumer threads, 4 rows each), prefix sum, table; the producer's staging descriptors ----
    int nv[4], x0s[4];
    int local = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = tid * 4 + r;
        int mn = 0, mx = -1;
        const bool used = (tid < 256) && (row < rows) && packed_row_span(geo, row / Ly, row % Ly, &mn, &mx);
        x0s[r] = used ? (mn & ~3) : 0;
        nv[r] = used ? (((mx - x0s[r]) >> 2) + 1) : 0;
        local += nv[r];
    }
    const int incl = wave_scan_add(local);
    if (lane == 63 && wave < 4) scratch[wave] = incl;
    __syncthreads();
    int wave_off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) wave_off += (w < wave) ? scratch[w] : 0;
    const int nvec = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    const int nvec_pad = (nvec + 63) & ~63;
    if (tid < 256) {
        int run = wave_off + incl - local;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            if (row < rows) { tbl[4 * row] = run; tbl[4 * row + 1] = x0s[r]; }
            run += nv[r];
        }
    }
    const bool fits = (nvec_pad * 4 <= p.Lx) && (nvec_pad <= 64 * kSpanWsMaxIt) && (nvec_pad <= 8 * 320);
    unsigned short* const vrow = reinterpret_cast<unsigned short*>(buf);
    if (fits && tid < 256) {
        int first = wave_off + incl - local;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            for (int i = 0; i < nv[r]; ++i) vrow[first + i] = (unsigned short)row;
            first += nv[r];
        }
    }
    __syncthreads();
    // The staging descriptors -- byte offset of vector v relative to the box origin, -1 behind the last vector -- live in LDS behind the two
    // buffers: the producer wave reads its share (lane l: vectors l, l + 64, ...) per tile, one conflict-free ds_read_b32 per vector, instead
    // of holding forty of them in registers the consumer waves would have to be allocated as well.
    int* const rel_l = reinterpret_cast<int*>(buf + 2 * p.Lx);
    int myrel[8];                                     // (computed before the barrier below: vrow overlays the buffers, rel_l does not)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int v = tid + 320 * i;
        const bool real = fits && v < nvec;
        const int row = real ? vrow[v] : 0;
        const int Z = row / Ly, Y = row - Z * Ly;
        const int xv = real ? tbl[4 * row + 1] + 4 * (v - tbl[4 * row]) : 0;
        myrel[i] = real ? ((Z * p.sH + Y) * p.sP + xv) * 4 : -1;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int v = tid + 320 * i;
        if (fits && v < nvec_pad) rel_l[v] = myrel[i];
    }
    __syncthreads();
    {
        v4i e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            e[r] = (v4i)(0);
            if (tid < 256 && row < rows) {
                auto addr = [&](int rr) { return (int)buf_b + 4 * (4 * tbl[4 * rr] - tbl[4 * rr + 1]); };
                const int y = row % Ly;
                const int r01 = (y + 1 < Ly) ? row + 1 : row, r10 = (row + Ly < rows) ? row + Ly : row;
                const int r11 = (y + 1 < Ly && row + Ly < rows) ? row + Ly + 1 : r10;
                e[r] = (v4i){addr(row), addr(r01), addr(r10), addr(r11)};
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tid * 4 + r;
            if (tid < 256 && row < rows) *reinterpret_cast<v4i*>(tbl + 4 * row) = e[r];
        }
    }

    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;
#ifdef VT_EXPERIMENTS      // make EXTRA=-DVT_EXPERIMENTS: VT_EXP_NOSTORE / VT_EXP_NOLOAD / VT_EXP_NOLDS ablations
    const bool no_stores = (p.flags & (1 << 21)) != 0, no_loads = (p.flags & (1 << 22)) != 0, no_lds = (p.flags & (1 << 26)) != 0;
#else
    constexpr bool no_stores = false, no_loads = false, no_lds = false;
#endif
    const bool plain_order = (p.flags & (1 << 23)) != 0;
    const int ntiles = p.nTd * p.nTh * p.nTw;
    const int nids = plain_order ? ntiles : blocked_tile_count(p.nTd, p.nTh, p.nTw);
    const int nSh = (p.nTh + 3) >> 2, nSw = (p.nTw + 3) >> 2;
    const int xcd = blockIdx.x & 7, per = (((nids + 63) >> 6) + 7) / 8 * 64;
    const int id0 = xcd * per, id_cnt = max(0, min(per, nids - id0));
    int* const counter = queue + 32 * xcd;
    const int64_t ostride = (int64_t)p.oH * p.oW;
    __syncthreads();                                  // the table is complete

    if (!fits) {
        // (never seen: the plan carries a margin over the host's own count) consumers gather their tiles from global memory
        if (tid == 0) scratch[0] = atomicAdd(counter, 1);
        __syncthreads();
        for (;;) {
            const int cur = scratch[0];
            __syncthreads();
            if (cur >= id_cnt) break;
            int td_i, th_i, tw_i;
            const bool tile_ok = plain_order ? (td_i = (id0 + cur) / (p.nTh * p.nTw), th_i = ((id0 + cur) / p.nTw) % p.nTh, tw_i = (id0 + cur) % p.nTw, true)
                                             : blocked_tile(id0 + cur, p.nTd, p.nTh, p.nTw, td_i, th_i, tw_i);
            if (tile_ok && tid < 256) {
                const int d0 = td_i * TD, h = th_i * TH + tid / TW, w = tw_i * TW + tid % TW;
                if (h < p.oH && w < p.oW) {
                    const int nd = min(TD, p.oD - d0);
#pragma unroll 1
                    for (int i = 0; i < nd; ++i) {
                        const int d = d0 + i;
                        float* optr = out + ((int64_t)d * p.oH + h) * p.oW + w;
                        if (canonical_inside(p, d, h, w)) {
                            double sc[3];
#pragma unroll
                            for (int r = 0; r < 3; ++r) sc[r] = canonical_coord(p, r, d, h, w);
                            const double fzd = floor(sc[0]), fyd = floor(sc[1]), fxd = floor(sc[2]);
                            *optr = direct_sample<0>(src, p, (int)fzd, (int)fyd, (int)fxd, (float)(sc[0] - fzd), (float)(sc[1] - fyd), (float)(sc[2] - fxd));
                        } else if (!keep) *optr = 0.0f;
                    }
                }
            }
            if (tid == 0) scratch[0] = atomicAdd(counter, 1);
            __syncthreads();
        }
    } else if (producer) {
        // =========================== producer wave ===========================
        // Tile parameters P (20 dwords, a ring of four slots): [0] mode (-1 none left, 0 no tile, 1 outside: zero-fill, 2 to be staged and
        // gathered) [1] flags (1 all_valid, 2 whole, 4 box inside the volume) [2..4] d0, h0, w0 [5..10] the tile's Q32.32 sub-voxel base
        // (lo, hi) x 3 [11..12] element offset of the tile in the output [13..14] element offset of the box origin in the source
        // [15..17] box origin o0, o1, o2.  P(k+2) is computed while the consumers stage tile k+1 and gather tile k.
        // The launch's float64 constants live in VECTOR registers of this wave (it has them to spare: the kernel's allocation is set by the
        // consumers): as scalars they are ~50 registers that get spilled, and re-read from the kernel-argument segment per tile they were 36
        // scalar loads behind 27 waits -- [measured] 5 800 cycles per tile with the consumers idle, the bound of the whole launch.
        double cm[12], cneg[3], cpos[3], cvlo[3], cvhi[3];
#pragma unroll
        for (int i = 0; i < 12; ++i) { cm[i] = p.m[i]; asm volatile("" : "+v"(cm[i])); }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            cneg[r] = p.neg[r]; cpos[r] = p.pos[r]; cvlo[r] = p.vlo[r]; cvhi[r] = p.vhi[r];
            asm volatile("" : "+v"(cneg[r]), "+v"(cpos[r]), "+v"(cvlo[r]), "+v"(cvhi[r]));
        }
        const int c_nTd = p.nTd, c_nTh = p.nTh, c_nTw = p.nTw, c_oD = p.oD, c_oH = p.oH, c_oW = p.oW, c_sD = p.sD, c_sH = p.sH, c_sP = p.sP, c_Lxbox = geo.Lxbox;
        const unsigned c_mw = p.nTw_magic, c_mh = p.nTh_magic;
        auto prepare = [&](int cur, int* P) {
            int mode = -1, flags = 0, d0 = 0, h0 = 0, w0 = 0, o0 = 0, o1 = 0, o2 = 0;
            uint64_t bfx[3] = {0, 0, 0};
            int64_t otile = 0, origin = 0;
            do {
                if (cur >= id_cnt) break;
                mode = 0;
                const int t = id0 + cur;
                int td_i, th_i, tw_i;
                bool tile_ok = true;
                if (plain_order) {
                    tw_i = t % c_nTw;
                    const int t2 = t / c_nTw;
                    th_i = t2 % c_nTh;
                    td_i = t2 / c_nTh;
                } else {
                    const unsigned sbi = (unsigned)t >> 6, l6 = (unsigned)t & 63u;
                    const unsigned s2 = nSw == 1 ? sbi : __umulhi(sbi, c_mw), sbw = sbi - s2 * (unsigned)nSw;
                    const unsigned sbd = nSh == 1 ? s2 : __umulhi(s2, c_mh), sbh = s2 - sbd * (unsigned)nSh;
                    td_i = (int)(sbd * 4 + (l6 >> 4)); th_i = (int)(sbh * 4 + ((l6 >> 2) & 3)); tw_i = (int)(sbw * 4 + (l6 & 3));
                    tile_ok = td_i < c_nTd && th_i < c_nTh && tw_i < c_nTw;
                }
                if (!tile_ok) break;
                d0 = td_i * TD; h0 = th_i * TH; w0 = tw_i * TW;
                double base[3], lo[3];
                bool any_valid = true, all_valid = true;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    base[r] = fma(cm[4 * r], (double)d0, fma(cm[4 * r + 1], (double)h0, fma(cm[4 * r + 2], (double)w0, cm[4 * r + 3])));
                    lo[r] = base[r] + cneg[r];
                    const double hi = base[r] + cpos[r];
                    any_valid = any_valid && (hi >= cvlo[r] - kTileMargin) && (lo[r] < cvhi[r] + kTileMargin);
                    all_valid = all_valid && (lo[r] >= cvlo[r] + kTileMargin) && (hi < cvhi[r] - kTileMargin);
                }
                const bool whole = (d0 + TD <= c_oD) && (h0 + TH <= c_oH) && (w0 + TW <= c_oW);
                flags = (all_valid ? 1 : 0) | (whole ? 2 : 0);
                otile = (int64_t)d0 * ostride + (int64_t)h0 * c_oW + w0;
                mode = 1;
                if (!any_valid) break;
                mode = 2;
                o0 = (int)floor(lo[0] - kSpanOriginMargin); o1 = (int)floor(lo[1] - kSpanOriginMargin); o2 = (int)floor(lo[2] - kSpanOriginMargin) & ~3;
                const bool box_inside = o0 >= 0 && o1 >= 0 && o2 >= 0 && o0 + Lz <= c_sD && o1 + Ly <= c_sH && o2 + c_Lxbox <= c_sP;
                flags |= box_inside ? 4 : 0;
                origin = ((int64_t)o0 * c_sH + o1) * c_sP + o2;
                bfx[0] = fx64(base[0] - (double)o0);
                bfx[1] = fx64(base[1] - (double)o1);
                bfx[2] = fx64(base[2] - (double)o2);
            } while (false);
            if (lane == 0) {
                *reinterpret_cast<v4i*>(P) = (v4i){mode, flags, d0, h0};
                *reinterpret_cast<v4i*>(P + 4) = (v4i){w0, (int)(uint32_t)bfx[0], (int)(uint32_t)(bfx[0] >> 32), (int)(uint32_t)bfx[1]};
                *reinterpret_cast<v4i*>(P + 8) = (v4i){(int)(uint32_t)(bfx[1] >> 32), (int)(uint32_t)bfx[2], (int)(uint32_t)(bfx[2] >> 32), (int)(uint32_t)(uint64_t)otile};
                *reinterpret_cast<v4i*>(P + 12) = (v4i){(int)(uint32_t)((uint64_t)otile >> 32), (int)(uint32_t)(uint64_t)origin, (int)(uint32_t)((uint64_t)origin >> 32), o0};
                *reinterpret_cast<v4i*>(P + 16) = (v4i){o1, o2, 0, 0};
            }
            return mode;
        };
        // ids: fetched by this wave alone, one per tile.  The atomic for id(k+3) is issued at the END of iteration k-1 and its value read at
        // the end of iteration k, a whole iteration later (the barrier paces the producer with the consumers): its latency is never waited
        // for.  The wave has no other vector-memory traffic, so the compiler's vmcnt(0) in front of the read is a wait for that atomic alone.
        int id_a = 0, id_b = 0, pend = 0;
        if (lane == 0) { id_a = atomicAdd(counter, 1); id_b = atomicAdd(counter, 1); }
        int next_id = __builtin_amdgcn_readfirstlane(id_b);
        int mode_k = prepare(__builtin_amdgcn_readfirstlane(id_a), slot);   // P(0)
        int mode_k1 = 0;                                                     // P(k+1)'s mode, known from iteration k-1 on
        if (lane == 0) pend = atomicAdd(counter, 1);                         // id(2)
        for (int k = -1;; ++k) {
            __syncthreads();                                                 // barrier k
            if (k >= 0 && mode_k < 0) break;
            const int m2 = prepare(next_id, slot + 20 * ((k + 2) & 3));     // P(k+2)
            next_id = __builtin_amdgcn_readfirstlane(pend);                  // id(k+3), fetched an iteration ago
            if (lane == 0) pend = atomicAdd(counter, 1);                     // id(k+4)
            if (k >= 0) mode_k = mode_k1;                                   // (at k = -1 mode_k stays P(0)'s: barrier 0 comes next)
            mode_k1 = m2;
        }
    } else {
        // =========================== consumer waves ===========================
        const int kw = tid % TW, jh = tid / TW;
        const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
        uint64_t coff[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) coff[r] = fx64(fma(p.m[4 * r + 1], (double)jh, p.m[4 * r + 2] * (double)kw));
        const int ob = (jh * p.oW + kw) * 4;
        const uint64_t inc0 = ((uint64_t)(uint32_t)p.inc_hi[0] << 32) | p.inc_lo[0];
        const uint64_t inc1 = ((uint64_t)(uint32_t)p.inc_hi[1] << 32) | p.inc_lo[1];
        const uint64_t inc2 = ((uint64_t)(uint32_t)p.inc_hi[2] << 32) | p.inc_lo[2];
        const int oplane_b = (int)(ostride * 4);
        const unsigned rl_b = lds_byte_address(reinterpret_cast<const float*>(rel_l)) + 4u * (unsigned)tid;
        for (int k = -1;; ++k) {
            __syncthreads();                                                 // barrier k: P(k+1) visible, S(k) landed everywhere, C(k-1) over everywhere
            // ---- stage tile k+1 into the other buffer (its gather is an iteration away) ----
            {
                const int* Pn = slot + 20 * ((k + 1) & 3);
                const v4i n0 = *reinterpret_cast<const v4i*>(Pn);
                if (__builtin_amdgcn_readfirstlane(n0.x) == 2) {
                    const v4i n3 = *reinterpret_cast<const v4i*>(Pn + 12), n4 = *reinterpret_cast<const v4i*>(Pn + 16);
                    const bool box_inside = (__builtin_amdgcn_readfirstlane(n0.y) & 4) != 0;
                    const int64_t origin = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(n3.z) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(n3.y));
                    const unsigned m0v = buf_b + (((k + 1) & 1) ? bdelta : 0u) + 16u * (unsigned)wave_first;
                    if (box_inside) {
                        // the box is inside the volume: one descriptor based at its origin, the vector's byte offset as the vector offset
                        const uint64_t sb = (uint64_t)(size_t)(src + origin);
                        const v4i rs = {(int)(uint32_t)sb, (int)((uint32_t)(sb >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
#pragma unroll
                        for (int it = 0; it < kSpanMaxIt; ++it) {
                            if (wave_first + 256 * it < nvec_pad && !no_loads) {           // wave-uniform
                                const int r4 = *reinterpret_cast<const __attribute__((address_space(3))) int*>((size_t)(rl_b + 1024u * it));
                                span_dma16(max(r4, 0), rs, m0v + 4096u * it);
                            }
                        }
                    } else {
                        // rim: per-vector bounds tests (the vector's (z, y, x) inside the box by dividing its offset by the source's plane and
                        // row sizes: the box is no larger than the volume in y and x, host-checked); outside vectors come from a block of zeros
                        const int o0 = __builtin_amdgcn_readfirstlane(n3.w), o1 = __builtin_amdgcn_readfirstlane(n4.x), o2 = __builtin_amdgcn_readfirstlane(n4.y);
                        const int row_f = p.sP, plane_f = p.sH * row_f;
                        const float inv_row = 1.0f / (float)row_f, inv_plane = 1.0f / (float)plane_f;
#pragma unroll 1
                        for (int it = 0; it < kSpanMaxIt; ++it) {
                            if (wave_first + 256 * it >= nvec_pad) break;
                            const int r4 = *reinterpret_cast<const __attribute__((address_space(3))) int*>((size_t)(rl_b + 1024u * it));
                            const int u = r4 >> 2;
                            int Z = (int)((float)u * inv_plane);
                            int rem = u - Z * plane_f;
                            if (rem < 0) { --Z; rem += plane_f; }
                            if (rem >= plane_f) { ++Z; rem -= plane_f; }
                            int Y = (int)((float)rem * inv_row);
                            int xv = rem - Y * row_f;
                            if (xv < 0) { --Y; xv += row_f; }
                            if (xv >= row_f) { ++Y; xv -= row_f; }
                            const int gz = o0 + Z, gy = o1 + Y, gx = o2 + xv;
                            const bool inb = r4 >= 0 && (unsigned)gz < (unsigned)p.sD && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sP;
                            const float* g = inb ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(src + origin) + r4) : zeros16;
                            span_dma16_global(g, m0v + 4096u * it);
                        }
                    }
                }
            }
            // ---- tile k ----
            bool counted = false;                                            // exactly TD stores were issued behind the staging loads above
            if (k >= 0) {
                const int* P = slot + 20 * (k & 3);
                const v4i q0 = *reinterpret_cast<const v4i*>(P), q1 = *reinterpret_cast<const v4i*>(P + 4), q2 = *reinterpret_cast<const v4i*>(P + 8);
                const int q3 = P[12];
                const int mode = __builtin_amdgcn_readfirstlane(q0.x);
                if (mode < 0) break;
                if (mode > 0) {
                    const int flags = __builtin_amdgcn_readfirstlane(q0.y);
                    const int d0 = __builtin_amdgcn_readfirstlane(q0.z), h0 = __builtin_amdgcn_readfirstlane(q0.w), w0 = __builtin_amdgcn_readfirstlane(q1.x);
                    const bool all_valid = (flags & 1) != 0, whole = (flags & 2) != 0;
                    const uint64_t otile = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(q3) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(q2.w);
                    __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out + (int64_t)otile), 0, 0x7fffffff, 0x00020000);
                    if (mode == 1) {
                        // the whole tile maps outside the valid interval: zero-fill (or leave untouched)
                        if (!keep && !no_stores) {
                            const int h = h0 + jh, w = w0 + kw;
                            if (whole || (h < p.oH && w < p.oW)) {
                                const int nd = min(TD, p.oD - d0);
                                for (int i = 0; i < nd; ++i) __builtin_amdgcn_raw_buffer_store_b32(0u, orsrc, ob, i * oplane_b, 0);
                            }
                        }
                    } else {
                        const uint64_t b0 = ((uint64_t)(uint32_t)q1.z << 32) | (uint32_t)q1.y, b1 = ((uint64_t)(uint32_t)q2.x << 32) | (uint32_t)q1.w,
                                       b2 = ((uint64_t)(uint32_t)q2.z << 32) | (uint32_t)q2.y;
                        const uint64_t c0 = b0 + coff[0], c1 = b1 + coff[1], c2 = b2 + coff[2];
                        const unsigned bd = (k & 1) ? bdelta : 0u;
                        if (all_valid && whole) {
                            span_gather_column<TD, false, (TD <= 8 ? TD / 4 : 1)>(c0, c1, c2, inc0, inc1, inc2, tbl_b, Ly, orsrc, ob, oplane_b, 0u, 0u, no_stores, no_lds, bd);
                            counted = !no_stores;
                        } else {
                            // rim: the store masks by the canonical float64 test (vt_device.h), then the same gather
                            unsigned a = 0u, z = 0u;
                            const int h = h0 + jh, w = w0 + kw;
                            if (h < p.oH && w < p.oW) {
                                const int nd = min(TD, p.oD - d0);
#pragma unroll 1
                                for (int i = 0; i < nd; ++i) {
                                    const bool inside = all_valid || canonical_inside(p, d0 + i, h, w);
                                    a |= inside ? (1u << i) : 0u;
                                    z |= (!inside && !keep) ? (1u << i) : 0u;
                                }
                            }
                            span_gather_column<TD, true>(c0, c1, c2, inc0, inc1, inc2, tbl_b, Ly, orsrc, ob, oplane_b, a, z, false, false, bd);
                        }
                    }
                }
            }
            // the staging loads of tile k+1 have landed (this wave's); the TD stores of a gather without masks were issued behind them and
            // may stay in flight (vmcnt counts in issue order)
            if (counted) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(TD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
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
typedef void (*span_fn)(const float*, float*, const float*, int*, const AffineParams, const PackGeom);
struct SpanCfg { int td, th, tw; };
static const SpanCfg kSpan[] = {
    {16, 8, 32},
    {16, 16, 16},
    {8, 8, 32},
    {8, 16, 32},
    {8, 8, 32},          // 4..6: wave-specialised (affine_span_ws): 320 threads, two footprint buffers
    {16, 8, 32},
    {4, 8, 32},
};
int span_config_count() { return (int)(sizeof(kSpan) / sizeof(kSpan[0])); }
void span_config(int idx, int* td, int* th, int* tw) { *td = kSpan[idx].td; *th = kSpan[idx].th; *tw = kSpan[idx].tw; }
int span_rows_max() { return kSpanRowsMax; }
int span_vectors_max() { return 256 * kSpanMaxIt; }

static span_fn span_entry(int cfg)
{
    switch (cfg) {
        case 0: return affine_span<16, 8, 32>;
        case 1: return affine_span<16, 16, 16>;
        case 2: return affine_span<8, 8, 32>;
        case 3: return affine_span<8, 16, 32>;
        case 4: return affine_span_ws<8, 8, 32>;
        case 5: return affine_span_ws<16, 8, 32>;
        default: return affine_span_ws<4, 8, 32>;
    }
}
bool span_config_pipelined(int idx) { return idx >= 4; }

hipError_t init_span_kernels()
{
    for (int cfg = 0; cfg < span_config_count(); ++cfg) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(span_entry(cfg)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_affine_span(int cfg, const float* src, float* out, const float* zeros16, int* queue,
                              const AffineParams& p, const PackGeom& geo, int grid, int lds_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL(span_entry(cfg), dim3(grid), dim3(span_config_pipelined(cfg) ? 320 : 256), lds_bytes, stream, src, out, zeros16, queue, p, geo);
    return hipGetLastError();
}

}  // namespace vt
