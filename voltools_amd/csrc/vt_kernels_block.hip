// vt_kernels_block.hip -- general matrices (true 3-D rotations, the reference's own benchmark protocol, tests/benchmark.py:52-54):
// 8 x 16 x 16 output tiles gathered from a bank-tuned LDS box by compact lane blocks (kernel id 9, gfx950).
//
// What bounded the bounding-box kernel (`affine_tiled`, vt_kernels_affine.hip) on 512^3 cubic rotations was instruction issue: 327
// wave-instructions per 64 voxels (234 VALU, 38-48 LDS, 40 SALU), SQ_ACTIVE_INST_ANY = 90 % of the run time
// (profiles/r02_general512_cubic_summary.json), with an LDS conflict degree of 2.1 beside it.  This kernel is the same algorithm with
// the instruction stream and the LDS image designed together:
//   * the LDS row stride RS is a template constant: the 16 tap rows x 3 eight-byte reads of a cubic voxel are immediates off
//     8 address registers (4 planes x {first two pairs, third pair}) instead of 32 computed addresses;
//   * the 64 taps are summed as 48 packed FMAs (v_pk_fma_f32) on the pairs as `ds_read_b64` returns them:
//     S_k += (wz_c * wy_b) * pair_k(c, b), k = 0..2, then the six parity-shifted x weights once per voxel;
//   * a 32-lane group of `ds_read_b64` is a 2 x 4 x 4 block of output voxels, not 2 rows of 16: its taps sit in a compact source
//     neighbourhood, and the plane stride (a run-time value: Ly * RS + pad) is chosen per matrix by the host's bank model so that
//     this neighbourhood spreads over the 32 bank pairs (tools/gather_b64_sim.py: conflict degree 1.6 instead of 2.2; measured 1.59);
//   * a workgroup serves a brick of 2 x 2 x 2 tiles and keeps the box-relative source offset of each of its staging vectors in
//     registers: staging a tile is one `buffer_load ... lds` per 16-byte vector and nothing else (box wholly inside the volume; the
//     checked path serves tiles at the border);
//   * coordinates step in Q32.32 between the eight voxels of a thread (Gray order: one increment per step), output offsets are a
//     per-thread register plus a per-step scalar.
// The arithmetic differs from `affine_tiled` only in the association of the 64-term sum (pairs, then x), i.e. by float32
// rounding of a convex combination; tests hold it to the same tolerance against the oracle.  The columns of the aligned 6-wide window
// that are not taps are dropped by selects (not by zero weights), so a non-finite source value reaches exactly the outputs whose stencil
// contains it (tests/test_gpu_ranges.py).
// Measured limits of the design (profiles/r02_ablate_block_*.txt): staging moves 36 bytes per voxel through
// the CU's vector-memory path (~70 GB/s per CU from L2); the gather itself is VALU-bound (~120 per 64 voxels).  A variant that
// staged per-plane footprint rectangles instead of the box (4.6 instead of 8.9 floats per voxel) was correct and slower: its
// per-tile set-up (one lane per plane cutting the rotated tile by a slab) and 65536 short workgroups cost more than the bytes saved.
#include "vt_internal.h"
#include "vt_device.h"

#include <algorithm>
#include <type_traits>

namespace vt {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kBlkTD = 8, kBlkTW = 16;         // tile depth and width; the height is a template parameter: 16 (two workgroups per CU) or 8 (four)
constexpr int block_max_it(int th) { return th == 16 ? 20 : 13; }      // staging vectors per thread: boxes up to 80 KiB / 52 KiB (four workgroups per CU up to 40 KiB, three beyond)

// (Inline asm and hazards: the compiler's hazard recogniser does not look into asm statements -- vt_kernels_rows.hip met that with a
//  v_readfirstlane_b32.  This statement is safe without wait states: its only register input is the address VGPR, written by plain
//  integer VALU instructions, and a VALU result feeding a later instruction's VGPR operand is interlocked by the hardware (the gfx950
//  software hazards concern VALU-written SGPR / VCC / EXEC / M0 read by lane-access, VMEM or LDS-DMA instructions, trans and
//  double-precision results read by the very next VALU, and v_readlane / v_writelane lane selects: none applies to a ds_read address);
//  the result is consumed only behind the counted s_waitcnt lgkmcnt of VT_BLK_WAIT, whose "+v" operands tie it to the sums.)
template <int OFF>
__device__ __forceinline__ void lds_read_b64(v2f& r, unsigned a)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
}

__device__ __forceinline__ v2f pk_fma(v2f a, float w, v2f c) { return __builtin_elementwise_fma(a, (v2f)(w), c); }

// bspline_weights (bspline.h:102-112) for one fraction, evaluated on the pair (f, 1 - f): the formulas for (w3, w0) and for (w1, w2) are
// the same expression in f and in 1 - f, so each pair is one packed instruction sequence -- operation for operation what
// vt_device.h::bspline_weights does per weight (same rounding), seven instructions per axis instead of thirteen.
__device__ __forceinline__ void bspline_weights_pk(float f, v2f& w30, v2f& w12)
{
    const v2f fg = {f, 1.0f - f};
    const v2f sq = fg * fg;
    w30 = ((v2f)(1.0f / 6.0f) * sq) * fg;                                        // {w3, w0}
    const v2f t = (v2f)(2.0f) - fg;
    w12 = (v2f)(2.0f / 3.0f) - ((v2f)(0.5f) * sq) * t;                            // {w1, w2} (contracted to one packed FMA, as the scalar form is)
}

// 64 taps of one voxel.  a0 = LDS byte address of column e (even) of tap row (z tap 0, y tap 0); ps4 = plane stride in bytes.
template <bool SIMPLE, int RS>
__device__ __forceinline__ float cubic_block_sample(unsigned a0, unsigned ps4, int par, float fz, float fy, float fx)
{
    constexpr int RS4 = RS * 4;
    float wx[4], wy[4], wz[4];
    if constexpr (SIMPLE) {
        cubic_weights<true>(fx, wx);
        cubic_weights<true>(fy, wy);
        cubic_weights<true>(fz, wz);
    } else {
        v2f a30, a12;
        bspline_weights_pk(fx, a30, a12); wx[0] = a30.y; wx[1] = a12.x; wx[2] = a12.y; wx[3] = a30.x;
        bspline_weights_pk(fy, a30, a12); wy[0] = a30.y; wy[1] = a12.x; wy[2] = a12.y; wy[3] = a30.x;
        bspline_weights_pk(fz, a30, a12); wz[0] = a30.y; wz[1] = a12.x; wz[2] = a12.y; wz[3] = a30.x;
    }
    // The six values e .. e+5 of a row are three aligned pairs at fixed offsets; the taps are e+par .. e+par+3, picked by the parity at
    // the end.  (par = 0 does not need the third pair: it is read all the same -- one address register per plane, every read an
    // immediate offset -- and never selected; the row stride always holds it, vt_plan.hip: lx_used.)
    unsigned a[4];
    a[0] = a0; a[1] = a0 + ps4; a[2] = a[1] + ps4; a[3] = a[2] + ps4;

    // eight batches of two tap rows (6 reads); two batches in flight
    v2f t[2][6];
#define VT_BLK_ISSUE(c, h, r)                                                                                                   \
    lds_read_b64<(2 * h) * RS4>(r[0], a[c]); lds_read_b64<(2 * h) * RS4 + 8>(r[1], a[c]); lds_read_b64<(2 * h) * RS4 + 16>(r[2], a[c]);     \
    lds_read_b64<(2 * h + 1) * RS4>(r[3], a[c]); lds_read_b64<(2 * h + 1) * RS4 + 8>(r[4], a[c]); lds_read_b64<(2 * h + 1) * RS4 + 16>(r[5], a[c]);
#define VT_BLK_WAIT(n, r) \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]));
#define VT_BLK_SUM(c, h, r)                                                                       \
    {                                                                                             \
        const float w0 = wz[c] * wy[2 * h], w1 = wz[c] * wy[2 * h + 1];                           \
        S0 = pk_fma(r[0], w0, S0); S1 = pk_fma(r[1], w0, S1); S2 = pk_fma(r[2], w0, S2);          \
        S0 = pk_fma(r[3], w1, S0); S1 = pk_fma(r[4], w1, S1); S2 = pk_fma(r[5], w1, S2);          \
    }
    v2f S0 = {0.f, 0.f}, S1 = {0.f, 0.f}, S2 = {0.f, 0.f};
#define VT_BLK_WAITN(r) VT_BLK_WAIT(6, r)
    VT_BLK_ISSUE(0, 0, t[0])
    VT_BLK_ISSUE(0, 1, t[1]) VT_BLK_WAITN(t[0]) VT_BLK_SUM(0, 0, t[0])
    VT_BLK_ISSUE(1, 0, t[0]) VT_BLK_WAITN(t[1]) VT_BLK_SUM(0, 1, t[1])
    VT_BLK_ISSUE(1, 1, t[1]) VT_BLK_WAITN(t[0]) VT_BLK_SUM(1, 0, t[0])
    VT_BLK_ISSUE(2, 0, t[0]) VT_BLK_WAITN(t[1]) VT_BLK_SUM(1, 1, t[1])
    VT_BLK_ISSUE(2, 1, t[1]) VT_BLK_WAITN(t[0]) VT_BLK_SUM(2, 0, t[0])
    VT_BLK_ISSUE(3, 0, t[0]) VT_BLK_WAITN(t[1]) VT_BLK_SUM(2, 1, t[1])
    VT_BLK_ISSUE(3, 1, t[1]) VT_BLK_WAITN(t[0]) VT_BLK_SUM(3, 0, t[0])
    VT_BLK_WAIT(0, t[1]) VT_BLK_SUM(3, 1, t[1])
#undef VT_BLK_WAITN
#undef VT_BLK_ISSUE
#undef VT_BLK_WAIT
#undef VT_BLK_SUM
    // the four columns of the stencil out of the six column sums, by selects: the extra columns of the aligned window never enter
    // the result (a zero weight would turn a non-finite neighbour, or whatever an earlier tile left in an unstaged slot, into NaN)
    const float t0 = par ? S0.y : S0.x, t1 = par ? S1.x : S0.y, t2 = par ? S1.y : S1.x, t3 = par ? S2.x : S1.y;
    return fmaf(wx[3], t3, fmaf(wx[2], t2, fmaf(wx[1], t1, wx[0] * t0)));
}

// Stage the box with per-vector bounds tests (tiles whose box leaves the volume: the rim, a third of the tiles of a rotated cube together
// with the tiles outside): vectors outside come from a block of zeros.  Round 4: the thread's own trimmed descriptors voff[] (0 = not
// staged) instead of the whole box, the vector's (z, y, x) by the set-up's multiply-high and a compile-time division instead of two
// run-time integer divisions: ~25 instead of ~90 instructions per vector, a third of the vectors.
// (the full-height kernel, 20 descriptors and 226 registers, keeps round 2's rolled loop over the whole box: the unrolled form spills there)
template <int RS>
__device__ __forceinline__ void stage_block_checked_box(float* lds, const float* __restrict__ src, const float* __restrict__ zeros16,
                                                        const AffineParams& p, const int (&o)[3], int total, int psv, int nvx_used, int tid)
{
    constexpr int nvx = RS / 4;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    for (int vb = wave_first; vb < total; vb += 256) {
        const int v = vb + (tid & 63);
        const int z = v / psv, rem = v - z * psv;
        const int y = rem / nvx, cx = rem - y * nvx;
        const int gz = o[0] + z, gy = o[1] + y, gx = o[2] + 4 * cx;
        const bool inb = (y < p.Ly) && (cx < nvx_used) && (unsigned)gz < (unsigned)p.sD && (unsigned)gy < (unsigned)p.sH &&
                         (unsigned)gx < (unsigned)p.sP;
        const float* g = inb ? src + (((int64_t)gz * p.sH + gy) * p.sP + gx) : zeros16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds + 4 * vb), 16, 0, 0);
    }
}

template <int RS, int MAXIT>
__device__ __forceinline__ void stage_block_checked(float* lds, const float* __restrict__ src, const float* __restrict__ zeros16,
                                                    const AffineParams& p, const int (&o)[3], const int (&voff)[MAXIT], int total, int psv, int tid)
{
    constexpr int nvx = RS / 4;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    const char* const origin = reinterpret_cast<const char*>(src + (((int64_t)o[0] * p.sH + o[1]) * p.sP + o[2]));
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        if (k * 256 + wave_first < total) {                      // wave-uniform
            int v = k * 256 + tid;
            asm volatile("" : "+v"(v));                          // (one vector at a time: the unrolled loop's address arithmetic, hoisted, spills)
            const int z = (int)__umulhi((unsigned)v, p.psv_magic), rem = v - z * psv;      // as in the set-up: v / psv
            const int y = rem / nvx, cx = rem - y * nvx;
            const int gz = o[0] + z, gy = o[1] + y, gx = o[2] + 4 * cx;
            const bool inb = (unsigned)gz < (unsigned)p.sD && (unsigned)gy < (unsigned)p.sH && (unsigned)gx < (unsigned)p.sP;
            const float* g = inb ? reinterpret_cast<const float*>(origin + voff[k]) : zeros16;
            if (voff[k] != 0 || v == 0)                          // (vector 0 is always staged: its offset is 0 too)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lds + 4 * (k * 256 + wave_first)), 16, 0, 0);
        }
    }
}

// TH = 16: 8 x 16 x 16 tiles, eight voxels per thread, boxes up to 80 KiB, two workgroups per CU (round 2's form).
// TH = 8 (round 4): 8 x 8 x 16 tiles, four voxels per thread, boxes up to 40 KiB and at most 128 registers: FOUR workgroups per CU.  The
// kernel's three phases -- tile set-up / voxel arithmetic, staging, LDS gather -- barely overlap inside one workgroup (ablations,
// profiles/r04_block_ablation.txt: 0.29 + 0.26 + 0.38 ms of a 1.03 ms launch at 512^3, and one workgroup per CU instead of two takes 1.53x
// as long): what hides a workgroup's staging latency is other workgroups' gathers, and four small ones interleave better than two large.
template <int KIND, int RS, int TH>
__device__ __forceinline__ void affine_block_body(const float* __restrict__ src, float* __restrict__ out, const float* __restrict__ zeros16,
                                                  int* __restrict__ queue, const AffineParams& p, const PackGeom& geo)
{
    constexpr bool CUBIC = KIND != 0;
    constexpr int HALO = CUBIC ? 1 : 0;
    constexpr int TD = kBlkTD, TW = kBlkTW;
    constexpr int NS = TH == 16 ? 8 : 4;                             // voxels per thread
    constexpr int kBlkMaxIt = block_max_it(TH);
    constexpr bool PIN = TH == 16;                                   // launch constants pinned in vector registers (plenty at 2 workgroups per CU)
    constexpr int nvx = RS / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    // a wave is a 4 x 4 x 4 block of voxels (32-lane groups 2 x 4 x 4), the workgroup 4 x 8 x 8; the eight voxels of a thread in
    // Gray order: w+8, h+8, w-8, d+4, w+8, h-8, w-8 (TH = 16), the four of the half-height tile: w+8, d+4, w-8
    const int vd = lane >> 4, vh = 4 * (wv >> 1) + ((lane >> 2) & 3), vw = 4 * (wv & 1) + (lane & 3);
    constexpr int sd[8] = {0, 0, TH == 16 ? 0 : 4, TH == 16 ? 0 : 4, 4, 4, 4, 4};
    constexpr int sh[8] = {0, 0, TH == 16 ? 8 : 0, TH == 16 ? 8 : 0, 8, 8, 0, 0};
    constexpr int sw[8] = {0, 8, 8, 0, 0, 8, 8, 0};
    constexpr int which[7] = {0, TH == 16 ? 1 : 3, 2, 3, 0, 4, 2};              // increments: 0 = +w8, 1 = +h8, 2 = -w8, 3 = +d4, 4 = -h8

    const int Lz = p.Lz, Ly = p.Ly, ps = p.Lps;
    const int psv = ps >> 2;
    const int total = Lz * psv;                                  // staging vectors (<= 256 * kBlkMaxIt, host-checked)
    const int nvx_used = p.Lx_used >> 2;
    const int row_b = p.sP * 4;
    const int plane_b = p.sH * row_b;                            // Lz * plane_b < 2^31 (host-checked)

    // box-relative byte offset of this thread's staging vectors; slots that hold no data (row / plane padding) re-read vector 0
    int voff[kBlkMaxIt];
#pragma unroll
    for (int k = 0; k < kBlkMaxIt; ++k) {
        const int v = k * 256 + tid;
        const int z = (int)__umulhi((unsigned)v, p.psv_magic), rem = v - z * psv;      // v / psv (v * psv < 2^32)
        const int y = rem / nvx, cx = rem - y * nvx;
        bool used = v < total && y < Ly && cx < nvx_used;
        if (used && (p.flags & (1 << 25))) {
            // Footprint trimming: of row (z, y) of the box only the columns some tile voxel can reach are staged -- the span that
            // packed_row_span (vt_internal.h) proves for EVERY sub-voxel position of a tile.  The LDS image keeps the box's strides, so
            // the gather is unchanged; unstaged slots keep whatever an earlier tile left there: the aligned 6-wide window of the cubic
            // gather may read such a slot, and drops it by a select.
            int mn, mx;
            used = packed_row_span(geo, z, y, &mn, &mx);
            used = used && cx >= (mn >> 2) && cx <= (mx >> 2);
        }
        voff[k] = used ? z * plane_b + y * row_b + 16 * cx : 0;
    }

    const bool keep = (p.flags & VT_KEEP_OUTSIDE) != 0;
    const int64_t ostride = (int64_t)p.oH * p.oW;
    const int obase = (int)((vd * ostride + (int64_t)vh * p.oW + vw) * 4);     // TD * ostride * 4 < 2^31 (host-checked)
    const int ostep_d = (int)(4 * ostride * 4), ostep_h = (int)(8 * (int64_t)p.oW * 4);
    const unsigned ps4 = 4u * (unsigned)ps;
    const unsigned lds_base = lds_byte_address(lds);
    // The step increments live in vector registers (there are plenty at 2 workgroups per CU): as scalars they are 30 of ~150 live
    // SGPR values of the tile loop, and the spilled ones come back through v_readlane at every use.
    int inc_hi[5][3];
    unsigned inc_lo[5][3];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            inc_hi[k][r] = p.binc_hi[k][r];
            inc_lo[k][r] = p.binc_lo[k][r];
            // (the half-height kernel walks w+8, d+4, w-8: kinds 0, 3, 2 -- 18 registers it can afford)
            if (PIN || k == 0 || k == 2 || k == 3) asm volatile("" : "+v"(inc_hi[k][r]), "+v"(inc_lo[k][r]));
        }
    // ... and so do the float64 constants of the tile geometry (float64 arithmetic is vector arithmetic anyway) where registers are
    // plentiful (PIN).  The half-height kernel (128 registers) reads them afresh for every tile instead, through a kernel-argument
    // pointer the compiler cannot see through: hoisted out of the tile loop these ~70 scalar registers stayed live across the voxel loop,
    // and the allocator's spills (v_readlane per use) landed on values the voxel loop reads -- 10 readlanes per voxel, ~100 per tile.
    double gm[12], gneg[3], gpos[3], gvlo[3], gvhi[3];
    double mp[3][3];                                              // matrix columns in the order of canonical_inside's chain (vt_device.h)
    const int oc0 = p.ord[0], oc1 = p.ord[1], oc2 = p.ord[2];
    auto load_consts = [&](const auto& kp_) {
        const auto* const kp = &kp_;
#pragma unroll
        for (int i = 0; i < 12; ++i) gm[i] = kp->m[i];
#pragma unroll
        for (int r = 0; r < 3; ++r) { gneg[r] = kp->neg[r]; gpos[r] = kp->pos[r]; gvlo[r] = kp->vlo[r]; gvhi[r] = kp->vhi[r]; }
    };
    // (the chain's columns: selects among the PINNED register copies of the matrix where it is pinned -- a select between two loads from
    // the argument struct becomes a select of two addresses, and the struct then lives on the stack: 528 bytes of scratch per thread --,
    // indexed loads where it is re-read per tile: there a select chain became fifteen scalar instructions per entry, per tile)
    auto load_chain = [&](const auto& kp_) {
        const auto* const kp = &kp_;
#pragma unroll
        for (int r = 0; r < 3; ++r) { mp[r][0] = kp->m[4 * r + oc0]; mp[r][1] = kp->m[4 * r + oc1]; mp[r][2] = kp->m[4 * r + oc2]; }
    };
    if constexpr (PIN) {
        load_consts(p);
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("" : "+v"(gm[i]));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            mp[r][0] = oc0 == 0 ? gm[4 * r] : (oc0 == 1 ? gm[4 * r + 1] : gm[4 * r + 2]);
            mp[r][1] = oc1 == 0 ? gm[4 * r] : (oc1 == 1 ? gm[4 * r + 1] : gm[4 * r + 2]);
            mp[r][2] = oc2 == 0 ? gm[4 * r] : (oc2 == 1 ? gm[4 * r + 1] : gm[4 * r + 2]);
            asm volatile("" : "+v"(gneg[r]), "+v"(gpos[r]), "+v"(gvlo[r]), "+v"(gvhi[r]), "+v"(mp[r][0]), "+v"(mp[r][1]), "+v"(mp[r][2]));
        }
    }
    double toff[3];                                               // M.(vd, vh, vw): this thread's first voxel relative to the tile's
#pragma unroll
    for (int r = 0; r < 3; ++r) toff[r] = fma(p.m[4 * r], (double)vd, fma(p.m[4 * r + 1], (double)vh, p.m[4 * r + 2] * (double)vw));
    auto inside_canonical = [&](int d, int h, int w) {
        const double x0 = (double)(oc0 == 0 ? d : (oc0 == 1 ? h : w));
        const double x1 = (double)(oc1 == 0 ? d : (oc1 == 1 ? h : w));
        const double x2 = (double)(oc2 == 0 ? d : (oc2 == 1 ? h : w));
        bool in = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double sc = fma(mp[r][0], x0, fma(mp[r][1], x1, fma(mp[r][2], x2, gm[4 * r + 3])));
            in = in && (sc >= gvlo[r]) && (sc < gvhi[r]);
        }
        return in;
    };
#ifdef VT_EXPERIMENTS      // make EXTRA=-DVT_EXPERIMENTS: VT_EXP_NOSTORE / VT_EXP_NOLOAD / VT_EXP_NOLDS ablations
    const bool no_stores = (p.flags & (1 << 21)) != 0, no_loads = (p.flags & (1 << 22)) != 0, no_lds = (p.flags & (1 << 26)) != 0;
#else
    constexpr bool no_stores = false, no_loads = false, no_lds = false;
#endif

    // Tiles are handed out from one counter per XCD (blockIdx % 8 is the XCD of a workgroup; each XCD owns a contiguous range of tile
    // ids, 4 x 4 x 4 tiles per super-block, w fastest): the tiles in flight on an XCD are always the most recent consecutive ids,
    // a compact patch of the volume whose overlapping boxes meet in that XCD's 4 MiB L2, while the workgroups stay persistent and
    // keep their staging offsets.  (Static striding frays the patch as soon as some workgroups meet cheap tiles outside the
    // volume -- L2 hit rate of the staging loads 0.62; bricks of 8 tiles per workgroup put 64 far-apart tiles in flight per XCD.)
    // The id of the next tile is fetched while the current one is gathered.  The last workgroup to leave zeroes the counters.
    int* const ctrl = reinterpret_cast<int*>(lds + (p.lds_cap >> 2));      // one word behind the box
    const int nids = blocked_tile_count(p.nTd, p.nTh, p.nTw);
    const int nSh = (p.nTh + 3) >> 2, nSw = (p.nTw + 3) >> 2;
    const int xcd = blockIdx.x & 7, per = ((nids >> 6) + 7) / 8 * 64;      // whole super-blocks per XCD
    const int id0 = xcd * per, id_cnt = max(0, min(per, nids - id0));
    // one counter per XCD, each on a cache line of its own; a fetch hands out `chunk` consecutive ids
    int* const counter = queue + 32 * xcd;
    const int chunk = p.dch;                                     // ids per fetch: 4 on large grids, fewer where a workgroup serves only a few tiles
    int nxt = 0;
    if (tid == 0) nxt = atomicAdd(counter, chunk);
    int cur = 0, left = 0;
    for (;;) {
        if (left == 0) {
            if (tid == 0) ctrl[0] = nxt;
            __syncthreads();                                     // the next chunk is visible
            cur = ctrl[0];
            left = chunk;
            __syncthreads();                                     // everyone has read it before thread 0 may overwrite it
            if (cur >= id_cnt) break;
            if (tid == 0) nxt = atomicAdd(counter, chunk);
        }
        const int id = id0 + cur;
        ++cur; --left;
        if (cur > id_cnt) continue;
        int td_i, th_i, tw_i;
        // blocked_tile (vt_device.h) with the two divisions as multiply-high by host constants (p.nTw_magic / p.nTh_magic hold the
        // magic numbers of the super-block counts along w / h here): the tile decode was half of the loop's uniform arithmetic
        const unsigned sbi = (unsigned)id >> 6, l6 = (unsigned)id & 63u;
        const unsigned s2 = nSw == 1 ? sbi : __umulhi(sbi, p.nTw_magic), sbw = sbi - s2 * (unsigned)nSw;      // (a count of 1 has no 32-bit magic number)
        const unsigned sbd = nSh == 1 ? s2 : __umulhi(s2, p.nTh_magic), sbh = s2 - sbd * (unsigned)nSh;
        td_i = (int)(sbd * 4 + (l6 >> 4)); th_i = (int)(sbh * 4 + ((l6 >> 2) & 3)); tw_i = (int)(sbw * 4 + (l6 & 3));
        const bool tile_ok = td_i < p.nTd && th_i < p.nTh && tw_i < p.nTw;
        if (!tile_ok) continue;
        if constexpr (!PIN) {
            // AffineParams follows the four pointer arguments of both kernels below (byte 32 of the kernel-argument segment)
            typedef const __attribute__((address_space(4))) char* KArg;
            KArg ka = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            load_consts(*(const __attribute__((address_space(4))) AffineParams*)(ka + 32));
        }
        const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;

        // ---- tile geometry (wave-uniform, float64) ----
        double base[3], lo[3], hi[3];
        bool any_valid = true, all_valid = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            base[r] = fma(gm[4 * r], (double)d0, fma(gm[4 * r + 1], (double)h0, fma(gm[4 * r + 2], (double)w0, gm[4 * r + 3])));
            lo[r] = base[r] + gneg[r];
            hi[r] = base[r] + gpos[r];
            any_valid = any_valid && (hi[r] >= gvlo[r] - kTileMargin) && (lo[r] < gvhi[r] + kTileMargin);
            all_valid = all_valid && (lo[r] >= gvlo[r] + kTileMargin) && (hi[r] < gvhi[r] - kTileMargin);
        }
        float* const otile = out + ((int64_t)d0 * ostride + (int64_t)h0 * p.oW + w0);
        __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(otile), 0, 0x7fffffff, 0x00020000);
        const bool whole = (d0 + TD <= p.oD) && (h0 + TH <= p.oH) && (w0 + TW <= p.oW);

        if (!any_valid) {
            // the whole tile maps outside the valid interval: zero-fill (or leave untouched)
            if (!keep) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int d = d0 + vd + sd[s], h = h0 + vh + sh[s], w = w0 + vw + sw[s];
                    if (whole || (d < p.oD && h < p.oH && w < p.oW))
                        __builtin_amdgcn_raw_buffer_store_b32(0u, orsrc, obase, (sd[s] >> 2) * ostep_d + (sh[s] >> 3) * ostep_h + 4 * sw[s], 0);
                }
            }
            continue;
        }

        int o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = (int)floor(lo[r]) - HALO;
        o[2] &= ~3;
        const bool box_inside = o[0] >= 0 && o[1] >= 0 && o[2] >= 0 && o[0] + Lz <= p.sD && o[1] + Ly <= p.sH && o[2] + p.Lx_used <= p.sP;

        __syncthreads();                                         // everyone is done gathering from the previous box
        if (no_loads) {
        } else if (box_inside) {
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(src + (((int64_t)o[0] * p.sH + o[1]) * p.sP + o[2]))), 0, 0x7fffffff, 0x00020000);
            char* dst = reinterpret_cast<char*>(lds) + 16 * wave_first;
#pragma unroll
            for (int k = 0; k < kBlkMaxIt; ++k)
                if (k * 256 + wave_first < total)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + 4096 * k), 16, voff[k], 0, 0, 0);
        } else {
            if constexpr (TH == 16) stage_block_checked_box<RS>(lds, src, zeros16, p, o, total, psv, nvx_used, tid);
            else stage_block_checked<RS, kBlkMaxIt>(lds, src, zeros16, p, o, voff, total, psv, tid);
        }
        // this thread's first voxel, box coordinates in Q32.32 (while the loads are in flight): the tile's base plus the thread's own
        // offset M.(vd, vh, vw), which is formed once per launch
        Fx c[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) c[r] = to_fx((base[r] - (double)o[r]) + toff[r]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- gather ----
        // Which of this thread's voxels store their value / a zero: all of them on tiles wholly inside the output and the valid interval
        // (all but the rim of the volume); on the others the canonical float64 test runs here, ahead of the gather, so that the voxel
        // loop below carries two bit tests per voxel instead of the tests' operands (15 of 210 instructions, formed for every voxel).
        unsigned inmask = (1u << NS) - 1u, zmask = 0u;
        if (!(all_valid && whole)) {
            inmask = 0u;
            if constexpr (!PIN) {
                typedef const __attribute__((address_space(4))) char* KArg;
                KArg ka = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(ka));
                load_chain(*(const __attribute__((address_space(4))) AffineParams*)(ka + 32));
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int d = d0 + vd + sd[s], h = h0 + vh + sh[s], w = w0 + vw + sw[s];
                if (d < p.oD && h < p.oH && w < p.oW) {
                    if (inside_canonical(d, h, w)) inmask |= 1u << s;
                    else if (!keep) zmask |= 1u << s;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float val;
            if (no_lds) {
                val = fx_frac(c[0]) + fx_frac(c[1]) + fx_frac(c[2]);
            } else if constexpr (CUBIC) {
                const int x1 = c[2].hi - 1, par = x1 & 1, e = x1 - par;
                const unsigned a0 = lds_base + 4u * (unsigned)(__mul24(c[0].hi - 1, ps) + __mul24(c[1].hi - 1, RS) + e);
                val = cubic_block_sample<KIND == 2, RS>(a0, ps4, par, fx_frac(c[0]), fx_frac(c[1]), fx_frac(c[2]));
            } else {
                const float* q = lds + (__mul24(c[0].hi, ps) + __mul24(c[1].hi, RS) + c[2].hi);
                const float* q1 = q + ps;
                const float fz = fx_frac(c[0]), fy = fx_frac(c[1]), fx = fx_frac(c[2]);
                const float a000 = q[0], a001 = q[1], a010 = q[RS], a011 = q[RS + 1];
                const float a100 = q1[0], a101 = q1[1], a110 = q1[RS], a111 = q1[RS + 1];
                const float x00 = fmaf(fx, a001 - a000, a000);
                const float x01 = fmaf(fx, a011 - a010, a010);
                const float x10 = fmaf(fx, a101 - a100, a100);
                const float x11 = fmaf(fx, a111 - a110, a110);
                const float y0 = fmaf(fy, x01 - x00, x00);
                const float y1 = fmaf(fy, x11 - x10, x10);
                val = fmaf(fz, y1 - y0, y0);
            }
            asm volatile("" : "+v"(val));                          // (this voxel's sums are complete before the next voxel's reads: see below)
            const int soff = (sd[s] >> 2) * ostep_d + (sh[s] >> 3) * ostep_h + 4 * sw[s];
            if (no_stores) {
                if (val == 123.456f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), orsrc, obase, soff, 0);
            } else if ((inmask >> s) & 1u) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), orsrc, obase, soff, 0);
            } else if ((zmask >> s) & 1u) {
                __builtin_amdgcn_raw_buffer_store_b32(0u, orsrc, obase, soff, 0);
            }
            if (s < NS - 1) {
#pragma unroll
                for (int r = 0; r < 3; ++r) fx_step(c[r], inc_hi[which[s]][r], inc_lo[which[s]][r]);
            }
            // One voxel at a time: this voxel's value and the next voxel's coordinates pass through volatile asm statements, which keep
            // their order among this voxel's reads and waits and the next one's.  Without them the loop body is one basic block, the
            // compiler lets one voxel's sums sink below the next voxel's 48 reads, needs 256 registers and spills 79.
            asm volatile("" : "+v"(c[0].hi), "+v"(c[0].lo), "+v"(c[1].hi), "+v"(c[1].lo), "+v"(c[2].hi), "+v"(c[2].lo));
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

// the two kernels: one body, two register / occupancy budgets (the second argument of __launch_bounds__ is waves per SIMD: 2 or 4
// workgroups of 256 threads per CU).  (Two entry points rather than a tile-height template parameter in the attribute: with a
// value-dependent __launch_bounds__ hipcc 7.2 emitted no host stub for kernels whose address is only taken through a function pointer.)
template <int KIND, int RS>
__global__ __launch_bounds__(256, 2) void affine_block(const float* __restrict__ src, float* __restrict__ out, const float* __restrict__ zeros16,
                                                        int* __restrict__ queue, const AffineParams p, const PackGeom geo)
{
    affine_block_body<KIND, RS, 16>(src, out, zeros16, queue, p, geo);
}
template <int KIND, int RS>
__global__ __launch_bounds__(256, 4) void affine_block_half(const float* __restrict__ src, float* __restrict__ out, const float* __restrict__ zeros16,
                                                             int* __restrict__ queue, const AffineParams p, const PackGeom geo)
{
    affine_block_body<KIND, RS, 8>(src, out, zeros16, queue, p, geo);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
typedef void (*block_fn)(const float*, float*, const float*, int*, const AffineParams, const PackGeom);

// row strides (floats).  Odd multiples of 4 are the bank-friendly ones (rows 8 apart share a bank pair); 16 / 24 / 32 serve boxes that
// exceed the staging budget at the next friendly stride.  The planner walks block_rs_order(th): 28, 36, 32 for the full-height tile (round
// 2's choice), every stride in order of preference for the half-height one.
static const int kBlkRS[] = {28, 36, 32, 12, 20, 16, 24};
static const int kOrder16[] = {0, 1, 2};
static const int kOrder8[] = {3, 4, 0, 1, 5, 6, 2};
int block_rs_count() { return (int)(sizeof(kBlkRS) / sizeof(kBlkRS[0])); }
int block_rs(int idx) { return kBlkRS[idx]; }
int block_rs_order(int th, const int** order) { *order = th == 16 ? kOrder16 : kOrder8; return th == 16 ? 3 : 7; }
int block_max_vectors(int th) { return 256 * block_max_it(th); }
void block_tile(int th, int* td, int* tw) { (void)th; *td = kBlkTD; *tw = kBlkTW; }

template <int RS, int TH>
static block_fn pick_block_kind(int kind)
{
    if constexpr (TH == 16) {
        switch (kind) {
            case 0: return affine_block<0, RS>;
            case 1: return affine_block<1, RS>;
            default: return affine_block<2, RS>;
        }
    } else {
        switch (kind) {
            case 0: return affine_block_half<0, RS>;
            case 1: return affine_block_half<1, RS>;
            default: return affine_block_half<2, RS>;
        }
    }
}

static block_fn block_entry(int rs_idx, int kind, int th)
{
    if (th == 16) return rs_idx == 0 ? pick_block_kind<28, 16>(kind) : (rs_idx == 1 ? pick_block_kind<36, 16>(kind) : pick_block_kind<32, 16>(kind));
    switch (rs_idx) {
        case 0: return pick_block_kind<28, 8>(kind);
        case 1: return pick_block_kind<36, 8>(kind);
        case 2: return pick_block_kind<32, 8>(kind);
        case 3: return pick_block_kind<12, 8>(kind);
        case 4: return pick_block_kind<20, 8>(kind);
        case 5: return pick_block_kind<16, 8>(kind);
        default: return pick_block_kind<24, 8>(kind);
    }
}

hipError_t init_block_kernels()
{
    for (int th = 8; th <= 16; th += 8)
        for (int rs = 0; rs < (th == 16 ? 3 : block_rs_count()); ++rs)
            for (int kind = 0; kind < 3; ++kind) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(block_entry(rs, kind, th)),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

hipError_t launch_affine_block(int rs_idx, int th, int interp, const float* src, float* out, const float* zeros16, int* queue,
                               const AffineParams& p, const PackGeom& geo, int grid, int lds_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL(block_entry(rs_idx, interp_kind(interp), th), dim3(grid), dim3(256), lds_bytes, stream, src, out, zeros16, queue, p, geo);
    return hipGetLastError();
}

}  // namespace vt
