// vt_kernels_prefilter.hip -- the separable cubic B-spline prefilter as three 1-D recursive passes.
//
// Reference: ConvertToInterpolationCoefficients (bspline.h:30-54) applied along X, Y, Z by
// SamplesToCoefficients3DX/Y/Z (bspline.h:58-99), one *thread per line*, each thread sweeping its line
// twice in global memory (causal, then anticausal).  On 512^3 that is 262144 threads with 2x512
// dependent steps each, an uncoalesced X pass, and every sample read and written twice per pass.
//
// Here the same recursion
//     c+[0] = L*(s[0] + sum_{n<min(12,N)} z^(n+1) s[n]);  c+[n] = L*s[n] + z*c+[n-1]
//     c[N-1] = z/(z-1) * c+[N-1];                         c[n]  = z*(c[n+1] - c+[n])
// (z = sqrt(3)-2, L = (1-z)(1-1/z)) is organised for a 64-wide wavefront machine:
//
//   X pass (contiguous lines): one wavefront per line, 64 consecutive samples per step held one per lane.
//     The first-order recursion is evaluated as a wave-level scan (log-step Hillis-Steele with the
//     ratio z^s, data moved with ds_bpermute via __shfl_up/__shfl_down), the carry between 64-sample
//     segments is a single readlane.  Loads and stores are fully coalesced, the whole line stays in
//     registers between the causal and the anticausal sweep: 4 B read + 4 B written per sample.
//
//   Y / Z passes (strided lines): lanes run along x (coalesced), and every lane owns a *chunk* of C
//     samples of one line.  Because |z|^16 = 7e-10, a chunk only needs K = 16 samples of warm-up before
//     it (causal) and after it (anticausal) to reproduce the full-line recursion to float32 precision;
//     the first and last chunk of a line use the reference's exact initialisations.  The chunk lives in
//     registers between the two sweeps, so HBM sees one read (plus warm-up overlap, absorbed by L2) and
//     one write per sample, and a 512^3 pass exposes 2M independent lanes instead of 262144.
//     Chunked passes read neighbours' samples, so they run out of place (ping-pong buffers).
#include "vt_internal.h"
#include "vt_device.h"

#include <cstdlib>

namespace vt {

__device__ __forceinline__ float zpow_small(int e)   // kPole^e for e >= 0 by binary decomposition (0 beyond 127)
{
    constexpr float z1 = kPole, z2 = z1 * z1, z4 = z2 * z2, z8 = z4 * z4, z16 = z8 * z8, z32 = z16 * z16, z64 = z32 * z32;
    float r = 1.0f;
    r *= (e & 1) ? z1 : 1.0f;
    r *= (e & 2) ? z2 : 1.0f;
    r *= (e & 4) ? z4 : 1.0f;
    r *= (e & 8) ? z8 : 1.0f;
    r *= (e & 16) ? z16 : 1.0f;
    r *= (e & 32) ? z32 : 1.0f;
    r *= (e & 64) ? z64 : 1.0f;
    return (e >= 128) ? 0.0f : r;                     // |z|^128 = 1e-73 underflows float32
}

// inclusive scan y[l] = t[l] + z*y[l-1] over the 64 lanes of a wave (y[-1] = 0)
__device__ __forceinline__ float wave_scan_up(float t, int lane)
{
    constexpr float z1 = kPole, z2 = z1 * z1, z4 = z2 * z2, z8 = z4 * z4, z16 = z8 * z8, z32 = z16 * z16;
    float u;
    u = __shfl_up(t, 1);  t = (lane >= 1)  ? fmaf(z1, u, t)  : t;
    u = __shfl_up(t, 2);  t = (lane >= 2)  ? fmaf(z2, u, t)  : t;
    u = __shfl_up(t, 4);  t = (lane >= 4)  ? fmaf(z4, u, t)  : t;
    u = __shfl_up(t, 8);  t = (lane >= 8)  ? fmaf(z8, u, t)  : t;
    u = __shfl_up(t, 16); t = (lane >= 16) ? fmaf(z16, u, t) : t;
    u = __shfl_up(t, 32); t = (lane >= 32) ? fmaf(z32, u, t) : t;
    return t;
}

// inclusive reverse scan y[l] = t[l] + z*y[l+1] (y[64] = 0)
__device__ __forceinline__ float wave_scan_down(float t, int lane)
{
    constexpr float z1 = kPole, z2 = z1 * z1, z4 = z2 * z2, z8 = z4 * z4, z16 = z8 * z8, z32 = z16 * z16;
    float u;
    u = __shfl_down(t, 1);  t = (lane < 63) ? fmaf(z1, u, t)  : t;
    u = __shfl_down(t, 2);  t = (lane < 62) ? fmaf(z2, u, t)  : t;
    u = __shfl_down(t, 4);  t = (lane < 60) ? fmaf(z4, u, t)  : t;
    u = __shfl_down(t, 8);  t = (lane < 56) ? fmaf(z8, u, t)  : t;
    u = __shfl_down(t, 16); t = (lane < 48) ? fmaf(z16, u, t) : t;
    u = __shfl_down(t, 32); t = (lane < 32) ? fmaf(z32, u, t) : t;
    return t;
}

__device__ __forceinline__ float wave_sum(float t)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) t += __shfl_xor(t, s);
    return t;
}

// ---- X pass: one wave per contiguous line, NSEG segments of 64 samples in registers ----
template <int NSEG>
__global__ __launch_bounds__(256) void prefilter_x_scan(const float* __restrict__ src, float* __restrict__ dst,
                                                         int W, int pitch, int64_t nlines, int lo_interior)
{
    const int lane = threadIdx.x & 63;
    const int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= nlines) return;                       // wave-uniform
    const float* s = src + line * pitch;
    float* o = dst + line * pitch;

    float v[NSEG];
#pragma unroll
    for (int g = 0; g < NSEG; ++g) {
        const int x = g * 64 + lane;
        v[g] = (x < W) ? s[x] : 0.0f;
    }

    const float zl1 = zpow_small(lane + 1);           // z^(lane+1): weight of the incoming carry
    // causal initialisation (bspline.h:2-19): s[0] + sum_{n < min(12,N)} z^(n+1) s[n]
    float init;
    {
        const int horizon = W < 12 ? W : 12;
        const float term = (lane < horizon) ? zl1 * v[0] : 0.0f;
        init = __shfl(v[0], 0) + wave_sum(term);
        if (lo_interior) init = __shfl(v[0], 0) * (1.0f / (1.0f - kPole));   // steady-state guess
    }
    float carry = 0.0f;
#pragma unroll
    for (int g = 0; g < NSEG; ++g) {
        float t = kLambda * v[g];
        if (g == 0) t = (lane == 0) ? kLambda * init : t;
        t = wave_scan_up(t, lane);
        t = fmaf(zl1, carry, t);
        carry = __shfl(t, 63);
        v[g] = t;                                     // c+
    }

    // anticausal: c[n] = u[n] + z*c[n+1], u[N-1] = z/(z-1)*c+[N-1], u[n<N-1] = -z*c+[n], u[n>=N] = 0
    const float zr = zpow_small(64 - lane);           // z^(64-lane): weight of the carry from the next segment
    carry = 0.0f;
#pragma unroll
    for (int g = NSEG - 1; g >= 0; --g) {
        const int x = g * 64 + lane;
        float u = (x < W - 1) ? (-kPole) * v[g] : ((x == W - 1) ? kAntiInit * v[g] : 0.0f);
        u = wave_scan_down(u, lane);
        u = fmaf(zr, carry, u);
        carry = __shfl(u, 0);
        if (x < W) o[x] = u;
    }
}

// ---- X pass, 16 bytes per lane: one wave per line, segments of 256 samples, 4 consecutive samples per lane ----
// Inside a lane the recursion is serial over its 4 samples; across lanes the carry obeys x -> L3 + z^4 x, a scan with
// ratio z^4 = 5.2e-3: three log-steps (z^4, z^8, z^16) reach float32 precision (the next term is z^32 = 5e-19).
template <int NSEG>
__global__ __launch_bounds__(256) void prefilter_x_scan4(const float* __restrict__ src, float* __restrict__ dst,
                                                          int W, int pitch, int64_t nlines, int lo_interior)
{
    constexpr float z1 = kPole, z2 = z1 * z1, z3 = z2 * z1, z4 = z2 * z2, z8 = z4 * z4, z16 = z8 * z8;
    const int lane = threadIdx.x & 63;
    const int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= nlines) return;                       // wave-uniform
    const float* s = src + line * pitch;
    float* o = dst + line * pitch;

    float4 v[NSEG];
#pragma unroll
    for (int g = 0; g < NSEG; ++g) {
        const int x = g * 256 + 4 * lane;
        v[g] = (x < W) ? *reinterpret_cast<const float4*>(s + x) : make_float4(0.f, 0.f, 0.f, 0.f);   // pitch pad is zero
    }
    const int lastx = W - 1;

    // causal initialisation (bspline.h:2-19): s[0] + sum_{n < min(12,N)} z^(n+1) s[n]; samples 0..11 sit in lanes 0..2
    float init;
    {
        const int horizon = W < 12 ? W : 12;
        const float zb = zpow_small(4 * lane + 1);
        float term = 0.f;
        if (4 * lane + 0 < horizon) term = fmaf(zb, v[0].x, term);
        if (4 * lane + 1 < horizon) term = fmaf(zb * z1, v[0].y, term);
        if (4 * lane + 2 < horizon) term = fmaf(zb * z2, v[0].z, term);
        if (4 * lane + 3 < horizon) term = fmaf(zb * z3, v[0].w, term);
        const float s0 = __shfl(v[0].x, 0);
        init = s0 + wave_sum(lane < 3 ? term : 0.f);
        if (lo_interior) init = s0 * (1.0f / (1.0f - kPole));
    }
    const float zl4 = zpow_small(4 * lane + 4);       // z^(4(lane+1)): weight of the segment carry at this lane's last sample
    float carry = 0.0f;                               // c+ of the last sample of the previous segment
#pragma unroll
    for (int g = 0; g < NSEG; ++g) {
        // local recursion with zero carry-in
        float l0 = kLambda * v[g].x;
        if (g == 0 && lane == 0) l0 = kLambda * init;           // c+[0] itself; the recursion starts at sample 1
        const float l1 = fmaf(z1, l0, kLambda * v[g].y);
        const float l2 = fmaf(z1, l1, kLambda * v[g].z);
        const float l3 = fmaf(z1, l2, kLambda * v[g].w);
        // last sample of every lane: t[l] = l3[l] + z^4 t[l-1]
        float t = l3, u;
        u = __shfl_up(t, 1); t = (lane >= 1) ? fmaf(z4, u, t) : t;
        u = __shfl_up(t, 2); t = (lane >= 2) ? fmaf(z8, u, t) : t;
        u = __shfl_up(t, 4); t = (lane >= 4) ? fmaf(z16, u, t) : t;
        t = fmaf(zl4, carry, t);
        float cin = __shfl_up(t, 1);
        cin = (lane == 0) ? carry : cin;
        if (g == 0 && lane == 0) cin = 0.f;                    // nothing precedes sample 0
        carry = __shfl(t, 63);
        v[g].x = fmaf(z1, cin, l0);
        v[g].y = fmaf(z2, cin, l1);
        v[g].z = fmaf(z3, cin, l2);
        v[g].w = fmaf(z4, cin, l3);
    }

    // anticausal: c[n] = u[n] + z c[n+1]; u[N-1] = z/(z-1) c+[N-1], u[n<N-1] = -z c+[n], u[n>=N] = 0
    const float zr4 = zpow_small(4 * (63 - lane) + 4);  // weight of the next segment's first sample at this lane's first sample
    carry = 0.0f;                                     // c of the first sample of the next segment
#pragma unroll
    for (int g = NSEG - 1; g >= 0; --g) {
        const int x = g * 256 + 4 * lane;
        auto uval = [&](int xx, float cp) { return (xx < lastx) ? (-kPole) * cp : ((xx == lastx) ? kAntiInit * cp : 0.0f); };
        const float r3 = uval(x + 3, v[g].w);
        const float r2 = fmaf(z1, r3, uval(x + 2, v[g].z));
        const float r1 = fmaf(z1, r2, uval(x + 1, v[g].y));
        const float r0 = fmaf(z1, r1, uval(x, v[g].x));
        float t = r0, u;
        u = __shfl_down(t, 1); t = (lane < 63) ? fmaf(z4, u, t) : t;
        u = __shfl_down(t, 2); t = (lane < 62) ? fmaf(z8, u, t) : t;
        u = __shfl_down(t, 4); t = (lane < 60) ? fmaf(z16, u, t) : t;
        t = fmaf(zr4, carry, t);
        float cin = __shfl_down(t, 1);
        cin = (lane == 63) ? carry : cin;
        carry = __shfl(t, 0);
        float4 c;
        c.w = fmaf(z1, cin, r3);
        c.z = fmaf(z2, cin, r2);
        c.y = fmaf(z3, cin, r1);
        c.x = fmaf(z4, cin, r0);
        if (x < W) {
            // samples >= W are pitch padding: u = 0 there, so c is z^k c[N-1]... no: c[n>=N] must stay 0
            if (x + 1 >= W) c.y = 0.f;
            if (x + 2 >= W) c.z = 0.f;
            if (x + 3 >= W) c.w = 0.f;
            *reinterpret_cast<float4*>(o + x) = c;
        }
    }
}

// ---- strided passes: one lane per (line, chunk), chunk + warm-up in registers ----
template <int C, int K>
__global__ __launch_bounds__(256) void prefilter_chunked(const float* __restrict__ src, float* __restrict__ dst,
                                                          int N, int64_t es,          // line length, element stride
                                                          int nA, int64_t sA,         // lane axis (coalesced)
                                                          int nB, int64_t sB,         // outer axis
                                                          int nchunks, int lo_interior, int chunk0)
{
    static_assert(C >= K && K >= 12, "chunk geometry");
    constexpr int R = C + 2 * K;
    const int lane = threadIdx.x & 63;
    const int nAb = (nA + 63) >> 6;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    // lane-axis blocks fastest, then the outer axis, then chunks ([measured] putting the chunks of one line set on
    // consecutive waves to share warm-up rows is 10 % slower: the 4 waves of a workgroup then stream 4 distant regions)
    const int ab = (int)(gw % nAb);
    const int64_t rest = gw / nAb;
    const int bi = (int)(rest % nB);
    const int chunk = chunk0 + (int)(rest / nB);     // chunk0: first chunk of this launch (axis-0 pass of the one-shot pipeline)
    if (chunk >= nchunks) return;                     // wave-uniform

    const int ai = ab * 64 + lane;
    const bool active = ai < nA;
    const int64_t lane_off = (int64_t)bi * sB + (int64_t)(active ? ai : nA - 1) * sA;
    const float* s = src + lane_off;
    float* o = dst + lane_off;

    const int a = chunk * C;                          // chunk covers [a, b)
    const int b = min(a + C, N);
    const int e = min(b + K, N);                      // anticausal sweep starts at e-1
    const int kl = e - 1 - a + K;                     // register index of position e-1

    float v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        int pos = a - K + k;
        pos = max(0, min(pos, N - 1));
        v[k] = s[(int64_t)pos * es];
    }

    // causal sweep
    float y;
    if (a == 0) {
        // positions < 0 do not exist: start at register K (position 0)
        if (!lo_interior) {
            float sum = v[K];
            float zn = kPole;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                if (n < N) sum += zn * v[K + n];
                zn *= kPole;
            }
            y = kLambda * sum;
        } else {
            y = kLambda * v[K] * (1.0f / (1.0f - kPole));
        }
        v[K] = y;
#pragma unroll
        for (int k = K + 1; k < R; ++k) { y = fmaf(kPole, y, kLambda * v[k]); v[k] = y; }
    } else {
        y = kLambda * v[0] * (1.0f / (1.0f - kPole));   // steady state of a constant signal; forgotten after K steps
        v[0] = y;
#pragma unroll
        for (int k = 1; k < R; ++k) { y = fmaf(kPole, y, kLambda * v[k]); v[k] = y; }
    }

    // anticausal sweep: the first valid position from the top uses c = z/(z-1)*c+ (exact at the line end,
    // bspline.h:27; the steady-state guess inside a line)
    float c = 0.0f;
#pragma unroll
    for (int k = R - 1; k >= K; --k) {
        c = (k == kl) ? kAntiInit * v[k] : kPole * (c - v[k]);
        v[k] = c;
    }

    if (active) {
        const int cnt = b - a;
#pragma unroll
        for (int k = 0; k < C; ++k)
            if (k < cnt) o[(int64_t)(a + k) * es] = v[K + k];
    }
}


// ---- strided passes, block form: a 1024-thread workgroup owns 256 columns x one segment of a line set, in registers ----
// The chunked kernel above loads 4 bytes per lane and re-reads 2 x 16 warm-up samples per 128-sample chunk (1.25-1.32x the
// samples: [measured] 3.4 TB/s of algorithmic traffic, 42 % of peak).  Here lanes run along x with 16 bytes each (a wave moves
// 1 KiB per instruction), the 16 waves of the workgroup split a segment of <= 16*CW consecutive samples of the line into
// chunks of CW, every wave runs the recursion on its chunk with zero carry-in, and the carries between chunks are combined
// EXACTLY through LDS: with g = z^CW the true carry into chunk w is sum_j g^(w-1-j) e_j over the local end states e_j of the
// chunks before it (Horner; the terms die out after one or two chunks but nothing is assumed).  Then y[k] += z^(k+1) carry.
// The anticausal sweep does the same from the other side.  Lines longer than one segment are cut into segments with
// K = 16 samples of warm-up at interior ends (|z|^16 = 7e-10), as the chunked kernel does for every chunk: 32 extra samples per
// 256 instead of per 128 -- and those re-reads meet the neighbouring segment's loads in L2.
// Same arithmetic per sample as bspline.h:30-54 (reference initialisations at true line ends), re-associated like the X pass.
constexpr int kBlkK = 16;           // warm-up samples at interior segment ends
// (waves, samples per wave): a workgroup covers NW*CW - 2*K samples of a line.  16 x 18 = 256 + 32 with one 1024-thread
// workgroup per CU; 8 x 20 = 128 + 32 with two 512-thread workgroups per CU (one loads while the other stores)

__device__ __forceinline__ float4 f4_fma(float a, const float4& b, const float4& c)
{
    return make_float4(fmaf(a, b.x, c.x), fmaf(a, b.y, c.y), fmaf(a, b.z, c.z), fmaf(a, b.w, c.w));
}
__device__ __forceinline__ float4 f4_scale(float a, const float4& b) { return make_float4(a * b.x, a * b.y, a * b.z, a * b.w); }

template <int NW, int kBlkCW>
__global__ __launch_bounds__(64 * NW, 4) void prefilter_block(const float* __restrict__ src, float* __restrict__ dst,
                                                          int N, int64_t es,          // line length, element stride along the line
                                                          int nA4,                    // columns / 4 (lane axis, contiguous, 16-byte vectors)
                                                          int nB, int64_t sB,         // outer axis
                                                          int nseg, int lo_interior, int W)
{
    constexpr int kBlkSeg = NW * kBlkCW - 2 * kBlkK;
    static_assert(kBlkCW >= 12, "the causal initialisation reads the first 12 samples from one chunk");
    __shared__ float4 ends[NW][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ncb = (nA4 + 63) >> 6;
    int t = blockIdx.x;
    const int cb = t % ncb; t /= ncb;
    const int bi = t % nB;
    const int seg = t / nB;                                   // segment of the line
    const int col4 = cb * 64 + lane;
    const bool active = col4 < nA4;
    const int64_t base = (int64_t)bi * sB + 4 * (int64_t)(active ? col4 : nA4 - 1);
    const float* s = src + base;
    float* o = dst + base;

    // positions of this workgroup: [a0, b0) are written; [la, lb) are loaded (warm-up at interior ends)
    const int a0 = seg * kBlkSeg, b0 = min(a0 + kBlkSeg, N);
    const int la = (seg > 0) ? a0 - kBlkK : 0;
    const int lb = min(b0 + ((b0 < N) ? kBlkK : 0), N);
    const int my0 = la + w * kBlkCW;                          // this wave's chunk [my0, my0 + CW) clipped to lb
    const int cnt = max(0, min(kBlkCW, lb - my0));

    float4 v[kBlkCW];
#pragma unroll
    for (int k = 0; k < kBlkCW; ++k)
        v[k] = (k < cnt) ? *reinterpret_cast<const float4*>(s + (int64_t)(my0 + k) * es) : make_float4(0.f, 0.f, 0.f, 0.f);
    // The pad columns W .. roundup4(W)-1 of the last vector are filtered along with the data and written to dst, where the
    // transform kernels read them as the border colour: whatever src holds there (a recycled ping-pong buffer), they enter as 0.
    // Block-uniform branch: only the last column block of a width that is not a multiple of 4 pays for the selects.
    if ((W & 3) && cb == ncb - 1) {
        const int x0 = 4 * col4;
#pragma unroll
        for (int k = 0; k < kBlkCW; ++k) {
            if (x0 + 1 >= W) v[k].y = 0.f;
            if (x0 + 2 >= W) v[k].z = 0.f;
            if (x0 + 3 >= W) v[k].w = 0.f;
        }
    }

    constexpr float z1 = kPole;
    float g = 1.0f;                                           // z^CW
#pragma unroll
    for (int k = 0; k < kBlkCW; ++k) g *= z1;

    // ---- causal: y[k] = L v[k] + z y[k-1] ----
    const bool line_start = (my0 == 0) && cnt > 0;            // this chunk holds sample 0 of the line (wave 0 of segment 0)
    if (line_start) {
        float4 init;
        if (!lo_interior) {
            // bspline.h:2-19: s[0] + sum_{n < min(12, N)} z^(n+1) s[n]   (kBlkCW >= 12: all inside this chunk)
            float4 sum = v[0];
            float zn = z1;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                if (n < N) sum = f4_fma(zn, v[n], sum);
                zn *= z1;
            }
            init = f4_scale(kLambda, sum);
        } else {
            init = f4_scale(kLambda * (1.0f / (1.0f - kPole)), v[0]);
        }
        v[0] = init;
    } else {
        // interior start: zero carry-in now, the true carry is added below.  The very first loaded sample of a segment that
        // starts inside the line begins from the steady state of a constant signal (forgotten after K samples).
        if (w == 0 && cnt > 0) v[0] = f4_scale(kLambda * (1.0f / (1.0f - kPole)), v[0]);
        else v[0] = f4_scale(kLambda, v[0]);
    }
#pragma unroll
    for (int k = 1; k < kBlkCW; ++k) v[k] = f4_fma(z1, v[k - 1], f4_scale(kLambda, v[k]));
    // end state of this chunk (the value at its last valid sample; later samples are padding: their "state" keeps decaying,
    // which is exactly what the carry formula wants when cnt == CW; a partial chunk is always the last one and feeds nobody)
    ends[w][lane] = v[kBlkCW - 1];
    __syncthreads();
    if (w > 0) {
        float4 c = ends[0][lane];
        for (int j = 1; j < w; ++j) c = f4_fma(g, c, ends[j][lane]);
        float zk = z1;
#pragma unroll
        for (int k = 0; k < kBlkCW; ++k) { v[k] = f4_fma(zk, c, v[k]); zk *= z1; }
    }
    __syncthreads();

    // ---- anticausal: c[n] = u[n] + z c[n+1];  u[N-1] = z/(z-1) c+[N-1], u[n < N-1] = -z c+[n] ----
    // local sweep with zero carry-in from above; positions >= lb are padding (u = 0)
    {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = kBlkCW - 1; k >= 0; --k) {
            const int pos = my0 + k;
            float4 u;
            if (k >= cnt) u = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (pos == N - 1) u = f4_scale(kAntiInit, v[k]);
            else if (pos == lb - 1) u = f4_scale(kAntiInit, v[k]);        // interior end of a segment: steady-state guess, forgotten after K samples
            else u = f4_scale(-z1, v[k]);
            c = f4_fma(z1, c, u);
            v[k] = c;
        }
    }
    ends[w][lane] = v[0];
    __syncthreads();
    if (w < NW - 1) {
        float4 c = ends[NW - 1][lane];
        for (int j = NW - 2; j > w; --j) c = f4_fma(g, c, ends[j][lane]);
        // v[k] += z^(CW - k) * carry  (carry = true value of the next chunk's first sample)
        float zk = z1;
#pragma unroll
        for (int k = kBlkCW - 1; k >= 0; --k) { v[k] = f4_fma(zk, c, v[k]); zk *= z1; }
    }

    if (active) {
#pragma unroll
        for (int k = 0; k < kBlkCW; ++k) {
            const int pos = my0 + k;
            if (k < cnt && pos >= a0 && pos < b0) *reinterpret_cast<float4*>(o + (int64_t)pos * es) = v[k];
        }
    }
}

// ---- X and Y passes in ONE kernel (round 3): a 1024-thread workgroup owns a 160-row x 512-column tile of one plane, in registers ----
// Three separate passes move 24 bytes per sample.  The X pass works along contiguous rows and the Y pass across them, so a tile
// that holds WHOLE rows (8 consecutive samples per lane: a wave covers 512 columns with two 16-byte loads per lane) can run the X
// recursion on each of its rows inside the wave that holds it -- serial over a lane's 8 samples, the carry between lanes a two-step
// scan with ratio z^8 = 2.7e-5 (the next term, z^32, is 5e-19) -- and then the Y recursion down its columns exactly as
// prefilter_block does: NW waves stacked along the line, CW rows each, local sweeps with zero carry-in, exact carries through LDS.
// The plane is read once and written once: 8 + 8 bytes per sample for two passes, plus the warm-up rows of interior row segments
// (16 on either side of 128: 1.25x reads, which meet the neighbouring segment's loads in L2 when they run together).  Rows wider
// than 512 are cut into column segments of 480 + 16 warm-up columns on interior sides.  Arithmetic per sample: the reference's
// recursion (bspline.h:30-54) with its initialisations at true line ends, re-associated like the other kernels here.
constexpr int kXyNW = 16, kXyCW = 10, kXyK = 16;
constexpr int kXyRows = kXyNW * kXyCW, kXyNetRows = kXyRows - 2 * kXyK;        // 160 loaded, 128 written
constexpr int kXyCols = 512, kXyNetCols = kXyCols - 2 * kXyK;                   // 512 loaded, 480 written (interior column segments)

template <int NW, int CW>
__global__ __launch_bounds__(64 * NW) void prefilter_xy(const float* __restrict__ src, float* __restrict__ dst,
                                                         int H, int W, int pitch, int64_t plane, int nsegY, int nsegX)
{
    constexpr float z1 = kPole, z2 = z1 * z1, z3 = z2 * z1, z4 = z2 * z2, z5 = z4 * z1, z6 = z4 * z2, z7 = z4 * z3, z8 = z4 * z4, z16 = z8 * z8;
    constexpr float zp[9] = {1.0f, z1, z2, z3, z4, z5, z6, z7, z8};
    __shared__ float ends[NW][8][64];
    // (the wave index as a SCALAR: row indices r0 + k and every test on them are then scalar too -- as vector values the compiler kept all
    // CW of them alive from the loads to the stores and spilled them: 11 scratch dwords per thread, written to HBM like any store)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // consecutive tiles (the row segments of one plane, then the next plane) go to ONE XCD: the segments of a plane are resident
    // together there and their overlapping warm-up rows meet in that XCD's L2
    int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int sx = t % nsegX; t /= nsegX;
    const int sy = t % nsegY;
    const int zpl = t / nsegY;

    // columns: [na, nb) are written, [la, lb) loaded; rows likewise
    const bool one_x = nsegX == 1, one_y = nsegY == 1;
    const int na = one_x ? 0 : sx * kXyNetCols, nb = one_x ? W : min(na + kXyNetCols, W);
    const int la = one_x ? 0 : max(na - kXyK, 0), lb = one_x ? W : min(nb + kXyK, W);
    const int ra = one_y ? 0 : sy * (NW * CW - 2 * kXyK), rb = one_y ? H : min(ra + (NW * CW - 2 * kXyK), H);
    const int lra = one_y ? 0 : max(ra - kXyK, 0), lrb = one_y ? H : min(rb + kXyK, H);
    const int r0 = lra + w * CW;                              // this wave's rows [r0, r0 + cnt)
    const int cnt = max(0, min(CW, lrb - r0));
    const int x0 = la + 8 * lane;                             // this lane's columns [x0, x0 + 8)
    const bool lane_on = x0 < lb;
    const float* s = src + (int64_t)zpl * plane + x0;
    float* o = dst + (int64_t)zpl * plane + x0;

    float v[CW][8];
#pragma unroll
    for (int k = 0; k < CW; ++k) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (k < cnt && lane_on) {
            const float* rp = s + (int64_t)(r0 + k) * pitch;
            a = *reinterpret_cast<const float4*>(rp);
            b = *reinterpret_cast<const float4*>(rp + 4);          // stays inside the row's pitch (host-checked: pitch >= roundup8(W))
        }
        v[k][0] = a.x; v[k][1] = a.y; v[k][2] = a.z; v[k][3] = a.w; v[k][4] = b.x; v[k][5] = b.y; v[k][6] = b.z; v[k][7] = b.w;
    }
    // samples at columns >= lb are not part of this segment's line (pitch padding, or the next segment's columns): they enter as 0
    if (x0 + 8 > lb) {
#pragma unroll
        for (int k = 0; k < CW; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (x0 + j >= lb) v[k][j] = 0.f;
    }

    // ================= X pass: every row of this wave, inside the wave =================
    const int lastx = lb - 1;                                 // the true line end, or an interior end (steady-state guess there)
    const int hx = W < 12 ? W : 12;                           // horizon of the causal initialisation
#pragma unroll
    for (int k = 0; k < CW; ++k) {
        float* a = v[k];
        // causal, local: l[j] = L a[j] + z l[j-1], zero carry-in; lane 0 starts from the initialisation
        float first = kLambda * a[0];
        if (la == 0) {
            // bspline.h:2-19: L (s[0] + sum_{n < min(12, N)} z^(n+1) s[n]); samples 0..7 in lane 0, 8..11 in lane 1
            float part = 0.f;
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < hx) part = fmaf(zp[j + 1], a[j], part);
            } else if (lane == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (8 + j < hx) part = fmaf(z8 * zp[j + 1], a[j], part);
            }
            const float tot = __shfl(part, 0) + __shfl(part, 1);
            if (lane == 0) first = kLambda * (a[0] + tot);
        } else if (lane == 0) {
            first = kLambda * (1.0f / (1.0f - kPole)) * a[0];  // interior start: steady state of a constant signal, forgotten after K samples
        }
        a[0] = first;
#pragma unroll
        for (int j = 1; j < 8; ++j) a[j] = fmaf(z1, a[j - 1], kLambda * a[j]);
        // true end state of every lane: E[l] = e[l] + z^8 E[l-1]
        float e = a[7], u;
        u = __shfl_up(e, 1); e = (lane >= 1) ? fmaf(z8, u, e) : e;
        u = __shfl_up(e, 2); e = (lane >= 2) ? fmaf(z16, u, e) : e;
        float cin = __shfl_up(e, 1);
        cin = (lane == 0) ? 0.f : cin;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaf(zp[j + 1], cin, a[j]);
        // anticausal: c[n] = u[n] + z c[n+1]; u[last] = z/(z-1) c+[last], u[n < last] = -z c+[n], u[n > last] = 0
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            const int x = x0 + j;
            const float uj = (x < lastx) ? (-z1) * a[j] : ((x == lastx) ? kAntiInit * a[j] : 0.f);
            a[j] = (j == 7) ? uj : fmaf(z1, a[j + 1], uj);
        }
        float st = a[0];
        u = __shfl_down(st, 1); st = (lane < 63) ? fmaf(z8, u, st) : st;
        u = __shfl_down(st, 2); st = (lane < 62) ? fmaf(z16, u, st) : st;
        float cr = __shfl_down(st, 1);
        cr = (lane == 63) ? 0.f : cr;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaf(zp[8 - j], cr, a[j]);
    }
    // (columns >= lb hold exact zeros again: u = 0 there and nothing flows in from beyond the line)

    // ================= Y pass: down the columns, CW rows per wave, exact carries through LDS =================
    float g = 1.0f;                                           // z^CW
#pragma unroll
    for (int k = 0; k < CW; ++k) g *= z1;
    const int hy = H < 12 ? H : 12;
    const bool y_start = (lra == 0);                          // this segment holds row 0 (in wave 0)
    if (y_start && hy > CW) {
        // the causal initialisation reads rows 0..11: rows CW..11 live in wave 1 -- hand their weighted sum over
        if (w == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float part = 0.f;
                float zn = g * z1;                            // z^(CW+1): weight of row CW
#pragma unroll
                for (int k = 0; k < 12 - CW; ++k) {
                    if (CW + k < hy) part = fmaf(zn, v[k][j], part);
                    zn *= z1;
                }
                ends[0][j][lane] = part;
            }
        }
        __syncthreads();
    }
    if (w == 0 && cnt > 0) {
        if (y_start) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float sum = v[0][j];
                float zn = z1;
#pragma unroll
                for (int k = 0; k < CW; ++k) {
                    if (k < hy) sum = fmaf(zn, v[k][j], sum);
                    zn *= z1;
                }
                if (hy > CW) sum += ends[0][j][lane];
                v[0][j] = kLambda * sum;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[0][j] = kLambda * (1.0f / (1.0f - kPole)) * v[0][j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[0][j] = kLambda * v[0][j];
    }
#pragma unroll
    for (int k = 1; k < CW; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[k][j] = fmaf(z1, v[k - 1][j], kLambda * v[k][j]);
    // (wave 0 overwrites ends[0] only after it has read the hand-over itself; no other wave writes ends[0])
#pragma unroll
    for (int j = 0; j < 8; ++j) ends[w][j][lane] = v[CW - 1][j];
    __syncthreads();
    if (w > 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float c = ends[0][j][lane];
            for (int i = 1; i < w; ++i) c = fmaf(g, c, ends[i][j][lane]);
            float zk = z1;
#pragma unroll
            for (int k = 0; k < CW; ++k) { v[k][j] = fmaf(zk, c, v[k][j]); zk *= z1; }
        }
    }
    __syncthreads();
    // anticausal, local: positions >= lrb are padding (u = 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float c = 0.f;
#pragma unroll
        for (int k = CW - 1; k >= 0; --k) {
            const int pos = r0 + k;
            float u;
            if (k >= cnt) u = 0.f;
            else if (pos == lrb - 1) u = kAntiInit * v[k][j];                 // the line's end, or an interior end (steady-state guess)
            else u = (-z1) * v[k][j];
            c = fmaf(z1, c, u);
            v[k][j] = c;
        }
        ends[w][j][lane] = v[0][j];
    }
    __syncthreads();
    if (w < NW - 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float c = ends[NW - 1][j][lane];
            for (int i = NW - 2; i > w; --i) c = fmaf(g, c, ends[i][j][lane]);
            float zk = z1;
#pragma unroll
            for (int k = CW - 1; k >= 0; --k) { v[k][j] = fmaf(zk, c, v[k][j]); zk *= z1; }
        }
    }

    // ---- store rows [ra, rb) x columns [na, nb); behind the line's end the vector is completed with zeros (the pitch padding) ----
    const int nb_st = (nb == W) ? ((W + 3) & ~3) : nb;
    const bool st0 = x0 >= na && x0 < nb_st, st1 = x0 + 4 >= na && x0 + 4 < nb_st;
    // The row addresses are rebuilt here from one opaque offset: left to itself the compiler forms all CW of them next to the load
    // addresses at the top of the kernel and keeps them alive to this point -- 11 spilled registers in a 128-register kernel, whose
    // scratch stores reach HBM like any other store (17 % more bytes written than the tile itself, profiles/r03_prefilter512_summary.json)
    int64_t roff = (int64_t)r0 * pitch;
    asm volatile("" : "+v"(roff));
    float* rp = o + roff;
#pragma unroll
    for (int k = 0; k < CW; ++k, rp += pitch) {
        const int pos = r0 + k;
        if (k < cnt && pos >= ra && pos < rb) {
            if (st0) *reinterpret_cast<float4*>(rp) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
            if (st1) *reinterpret_cast<float4*>(rp + 4) = make_float4(v[k][4], v[k][5], v[k][6], v[k][7]);
        }
    }
}

// X + Y in one launch where the tile shape pays: rows of >= 64 samples, lines of >= 40; 16-byte aligned rows whose pitch holds whole
// 8-sample lanes.  Returns false when the shape does not qualify (the caller runs the two passes separately).
bool prefilter_xy_ok(int D, int H, int W, int pitch, const void* src, const void* dst)
{
    static const bool off = getenv("VT_PF_NO_XY") != nullptr;
    // (a caller's offset view through vt_prefilter_inplace may be 4-byte aligned only: the separate passes take it)
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0 || src == dst) return false;
    return !off && W >= 64 && H >= 40 && (pitch & 3) == 0 && pitch >= ((W + 7) & ~7) && (int64_t)D * ((H + 127) / 128) * ((W + 479) / 480) < 0x7fffffffLL;
}

hipError_t launch_prefilter_xy(const float* src, float* dst, int D, int H, int W, int pitch, hipStream_t stream)
{
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0 || src == dst) return hipErrorInvalidValue;
    const int nsegY = (H <= kXyRows) ? 1 : (H + kXyNetRows - 1) / kXyNetRows;
    const int nsegX = (W <= kXyCols) ? 1 : (W + kXyNetCols - 1) / kXyNetCols;
    const int64_t blocks = (int64_t)D * nsegY * nsegX;
    hipLaunchKernelGGL((prefilter_xy<kXyNW, kXyCW>), dim3((unsigned)blocks), dim3(64 * kXyNW), 0, stream,
                       src, dst, H, W, pitch, (int64_t)H * pitch, nsegY, nsegX);
    return hipGetLastError();
}

constexpr int kChunk = 64, kWarm = 16;
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// [measured] 128-sample chunks (1.25x read overlap, 160 data registers): 512^3 0.82 vs 0.85 ms, 1024^3 6.4 vs 6.95 ms
int prefilter_chunk_size(int N)
{
    const int variant = env_int("VT_PF_CHUNK", N >= 256 ? 128 : 64);
    return (variant == 128) ? 128 : (variant == 32 ? 32 : kChunk);
}

// chunks [c0, c1) of every line (all of them for the ordinary passes; a range for the axis-0 pass of the one-shot pipeline,
// which filters a chunk as soon as the planes it reads -- its own and kWarm on each side -- are there; same chunk grid,
// same bits)
static hipError_t launch_chunked(int C, const float* src, float* dst, int N, int64_t es, int nA, int64_t sA, int nB, int64_t sB,
                                 int c0, int c1, bool lo_interior, hipStream_t stream)
{
    const int nchunks = (N + C - 1) / C;
    c1 = c1 < nchunks ? c1 : nchunks;
    if (c0 < 0 || c0 >= c1) return hipSuccess;
    const int64_t waves = (int64_t)((nA + 63) / 64) * nB * (c1 - c0);
    const int64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    // the kernel bounds its chunk index by `nchunks`: pass the end of the range
    if (C == 128)
        hipLaunchKernelGGL((prefilter_chunked<128, kWarm>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           src, dst, N, es, nA, sA, nB, sB, c1, lo_interior ? 1 : 0, c0);
    else if (C == 32)
        hipLaunchKernelGGL((prefilter_chunked<32, kWarm>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           src, dst, N, es, nA, sA, nB, sB, c1, lo_interior ? 1 : 0, c0);
    else
        hipLaunchKernelGGL((prefilter_chunked<kChunk, kWarm>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           src, dst, N, es, nA, sA, nB, sB, c1, lo_interior ? 1 : 0, c0);
    return hipGetLastError();
}

int prefilter_warmup() { return kWarm; }

bool prefilter_axis_in_place_ok(int axis, int D, int H, int W)
{
    if (axis == 2) return W <= 2048;                  // whole line in registers before any store
    const int N = axis == 0 ? D : H;
    return N <= 32;                                   // single chunk in every variant: a lane reads its whole line first
}

hipError_t launch_prefilter_axis(int axis, const float* src, float* dst, int D, int H, int W, int pitch,
                                 bool lo_interior, hipStream_t stream)
{
    const int64_t plane = (int64_t)H * pitch;
    if (axis == 2 && W <= 2048 && (pitch & 3) == 0 && pitch >= ((W + 3) & ~3) &&
        ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        // rows are 16-byte aligned and padded to a multiple of 4 floats (the resident layout): 4 samples per lane
        const int64_t nlines = (int64_t)D * H;
        const int64_t blocks = (nlines + 3) / 4;
        if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
        const int nseg = (W + 255) / 256;
        const dim3 g((unsigned)blocks), b(256);
        const int li = lo_interior ? 1 : 0;
        if (nseg <= 1) hipLaunchKernelGGL(prefilter_x_scan4<1>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        else if (nseg <= 2) hipLaunchKernelGGL(prefilter_x_scan4<2>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        else if (nseg <= 4) hipLaunchKernelGGL(prefilter_x_scan4<4>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        else hipLaunchKernelGGL(prefilter_x_scan4<8>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        return hipGetLastError();
    }
    if (axis == 2 && W <= 2048) {
        const int64_t nlines = (int64_t)D * H;
        const int64_t blocks = (nlines + 3) / 4;
        if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
        const int nseg = (W + 63) / 64;
        const dim3 g((unsigned)blocks), b(256);
        const int li = lo_interior ? 1 : 0;
        if (nseg <= 1) hipLaunchKernelGGL(prefilter_x_scan<1>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        else if (nseg <= 2) hipLaunchKernelGGL(prefilter_x_scan<2>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        else if (nseg <= 4) hipLaunchKernelGGL(prefilter_x_scan<4>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        else if (nseg <= 8) hipLaunchKernelGGL(prefilter_x_scan<8>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        else if (nseg <= 16) hipLaunchKernelGGL(prefilter_x_scan<16>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        else hipLaunchKernelGGL(prefilter_x_scan<32>, g, b, 0, stream, src, dst, W, pitch, nlines, li);
        return hipGetLastError();
    }
    static const bool no_block = getenv("VT_PF_NO_BLOCK") != nullptr;
    static const int blk_variant = env_int("VT_PF_BLOCK", 0);
    if (!no_block && axis != 2 && src != dst && (pitch & 3) == 0 && pitch >= ((W + 3) & ~3) &&
        ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 && (axis == 1 ? H : D) >= 40) {
        // block form (16 bytes per lane, exact carries): rows are 16-byte aligned and padded to whole vectors (the pad columns
        // of the last vector enter as zeros whatever src holds there, so dst's pad columns are zeros)
        const int N = axis == 1 ? H : D;
        const int64_t es = axis == 1 ? (int64_t)pitch : plane;
        const int nB = axis == 1 ? D : H;
        const int64_t sB = axis == 1 ? plane : (int64_t)pitch;
        const int nA4 = ((W + 3) & ~3) / 4;
        auto launch = [&](auto kern, int nw, int cw) -> hipError_t {
            const int seg = nw * cw - 2 * kBlkK;
            const int nseg = (N + seg - 1) / seg;
            const int64_t blocks = (int64_t)((nA4 + 63) / 64) * nB * nseg;
            if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * nw), 0, stream, src, dst, N, es, nA4, nB, sB, nseg, lo_interior ? 1 : 0, W);
            return hipGetLastError();
        };
        if (blk_variant == 1) return launch(prefilter_block<8, 20>, 8, 20);
        if (blk_variant == 2) return launch(prefilter_block<8, 18>, 8, 18);
        return launch(prefilter_block<16, 18>, 16, 18);
    }
    int N, nA, nB;
    int64_t es, sA, sB;
    if (axis == 2) { N = W; es = 1; nA = H; sA = pitch; nB = D; sB = plane; }      // very wide lines: lanes along y
    else if (axis == 1) { N = H; es = pitch; nA = W; sA = 1; nB = D; sB = plane; }
    else { N = D; es = plane; nA = W; sA = 1; nB = H; sB = pitch; }
    const int C = prefilter_chunk_size(N);
    return launch_chunked(C, src, dst, N, es, nA, sA, nB, sB, 0, (N + C - 1) / C, lo_interior, stream);
}

hipError_t launch_prefilter_axis0_chunks(const float* src, float* dst, int D, int H, int W, int pitch, int c0, int c1, hipStream_t stream)
{
    const int C = prefilter_chunk_size(D);
    return launch_chunked(C, src, dst, D, (int64_t)H * pitch, W, 1, H, pitch, c0, c1, false, stream);
}

}  // namespace vt
