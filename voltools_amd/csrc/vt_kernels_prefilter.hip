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
