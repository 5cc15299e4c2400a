// Why do deep marching chunks write slower than short-lived workgroups (pattern_probe: 16 x 32 tiles, 4 B per lane: 6.4 TB/s at
// 4 planes per workgroup, 5.6 at 64, 5.4 at 256)?  Hypothesis: workgroups that march for long drift apart, the chip-wide write
// front gets ragged (128-byte pieces of many planes in flight at once) and DRAM page locality goes.  Test: the marching kernels'
// structure (tile marching through planes, one quad = 4 planes per step, 2-slot LDS ring fed by buffer_load ... lds from the
// plane-quad layout, barrier per step, counted waits) as a pure copy, with an optional PACING rule: the workgroups of one
// tile row (the ones that complete whole output rows together) may not run more than `delta` steps ahead of the slowest
// member -- one atomic add and one bounded poll per workgroup and step, no other communication.
//   mode 1 = loads only, 2 = stores only, 3 = both
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/front_probe.hip -o gpurun_out/front_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int xcd_contiguous(int b, int n)
{
    const int xcd = b & 7, q = n >> 3, r = n & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (b >> 3);
}

struct Args {
    const float* in; float* out; unsigned* cnt;
    int N, rowbytes, nsteps, nTh, nTw, mode, delta, gsz, expect_per_gen, gen, spin_max;
};

template <int TH, int TW, int NT>
__global__ __launch_bounds__(NT) void march_copy(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RP = NT / TW, NPIX = TH / RP, NSTORE = 4 * NPIX;
    const int tid = threadIdx.x;
    const int chunk = blockIdx.y;
    const int u = xcd_contiguous(blockIdx.x, gridDim.x);
    const int th_i = u / a.nTw, tw_i = u - th_i * a.nTw;
    const int N = a.N;
    const bool do_ld = a.mode & 1, do_st = a.mode & 2;
    const int rows = TH + 1, vpr = TW + 2, nvec = rows * vpr, nvec64 = (nvec + 63) & ~63, slot_bytes = nvec64 * 16;
    int voff[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int v = tid + NT * it;
        if (v >= nvec) v = 0;
        const int y = v / vpr, cx = v - y * vpr;
        int gy = th_i * TH + y, gx = (tw_i * TW + cx) * 16;
        if (gy >= N) gy = N - 1;
        if (gx + 16 > a.rowbytes) gx = 0;
        voff[it] = gy * a.rowbytes + gx;
    }
    const int nit = (nvec64 + NT - 1) / NT;
    const int step_bytes = N * a.rowbytes;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.in) + (size_t)chunk * a.nsteps * step_bytes), 0, 0x7fffffff, 0x00020000);
    const int kw = tid % TW, jh0 = tid / TW;
    const size_t plane = (size_t)N * N;
    __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(a.out + (size_t)chunk * a.nsteps * 4 * plane + (size_t)(th_i * TH) * N + tw_i * TW), 0, 0x7fffffff, 0x00020000);
    int ob[NPIX], q[NPIX];
#pragma unroll
    for (int px = 0; px < NPIX; ++px) { ob[px] = ((jh0 + px * RP) * N + kw) * 4; q[px] = ((jh0 + px * RP) * vpr + kw) * 16; }
    const int pb = N * N * 4;
    char* lds_c = reinterpret_cast<char*>(lds);
    auto issue = [&](int s, int slot_off) {
        char* dst = lds_c + slot_off + 16 * wave_first;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (it < nit && wave_first + NT * it < nvec64)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + 16 * NT * it), 16, voff[it], s * step_bytes, 0, 0);
    };
    // pacing group: gsz consecutive tiles of one tile row of one chunk
    const int ngrp = (a.nTw + a.gsz - 1) / a.gsz;
    const int grp = (chunk * a.nTh + th_i) * ngrp + tw_i / a.gsz;
    unsigned* mycnt = a.cnt + (size_t)grp * a.nsteps;
    const int members = min(a.gsz, a.nTw - (tw_i / a.gsz) * a.gsz);
    const unsigned expect = (unsigned)(a.gen - 1) * members + members;

    int slot = 0;
    if (do_ld) issue(0, 0);
    for (int s = 0; s < a.nsteps; ++s) {
        if (a.delta > 0 && s >= a.delta && tid == 0) {
            int n = 0;
            while (__hip_atomic_load(mycnt + (s - a.delta), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect && n < a.spin_max) { __builtin_amdgcn_s_sleep(4); ++n; }
        }
        if (do_ld) {
            if (s > 0 && do_st) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (do_ld && s + 1 < a.nsteps) issue(s + 1, slot ^ slot_bytes);
        if (do_st) {
            v4f part[NPIX];
#pragma unroll
            for (int px = 0; px < NPIX; ++px) {
                if (do_ld) part[px] = *reinterpret_cast<const v4f*>(lds_c + slot + q[px]);
                else { const v4f c = {1.f, 2.f, 3.f, (float)s}; part[px] = c; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int px = 0; px < NPIX; ++px)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, part[px][i]), orsrc, ob[px], (4 * s + i) * pb, 2);
        }
        if (a.delta > 0 && tid == 0) __hip_atomic_fetch_add(mycnt + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        slot ^= slot_bytes;
    }
}

template <int TH, int TW, int NT>
static void run(const float* in, float* out, unsigned* cnt, int N, int dch, int mode, int delta, int gsz, int lds_pad = 0)
{
    static int gen_of[64] = {0};
    const int Wq = (N + 1 + 7) & ~7;
    Args a;
    a.in = in; a.out = out; a.cnt = cnt; a.N = N; a.rowbytes = Wq * 16; a.nsteps = dch / 4; a.nTh = N / TH; a.nTw = N / TW;
    a.mode = mode; a.delta = delta; a.gsz = gsz; a.spin_max = 3000; a.gen = 0;
    const int nvec64 = ((TH + 1) * (TW + 2) + 63) & ~63;
    const int ldsb = 2 * nvec64 * 16 + lds_pad;
    void (*fn)(const Args) = march_copy<TH, TW, NT>;
    CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const dim3 grid(a.nTh * a.nTw, N / dch);
    CK(hipMemset(cnt, 0, (size_t)64 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int gen = 0;
    float r[5];
    for (int rep = -1; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) { a.gen = ++gen; hipLaunchKernelGGL(fn, grid, dim3(NT), ldsb, 0, a); }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        if (rep >= 0) CK(hipEventElapsedTime(&r[rep], e0, e1));
    }
    CK(hipGetLastError());
    for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (r[j] < r[i]) { float t = r[i]; r[i] = r[j]; r[j] = t; }
    const float ms = r[2] / 10;
    const double bytes = 4.0 * N * N * N * ((mode & 1 ? 1 : 0) + (mode & 2 ? 1 : 0));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)fn, NT, ldsb));
    printf("  %s %2dx%-2d nt=%d dch=%4d delta=%d gsz=%2d lds=%6d wg/cu=%d grid=%5dx%-3d : %.4f ms  %.2f TB/s\n", mode == 1 ? "R  " : mode == 2 ? "W  " : "R+W", TH, TW, NT, dch,
           delta, gsz, ldsb, occ, grid.x, grid.y, ms, bytes / ms / 1e9);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    (void)gen_of;
}

int main(int argc, char** argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : 0;
    for (int N : {512, 1024}) {
        if (only && N != only) continue;
        const size_t n = (size_t)N * N * N;
        const int Wq = (N + 1 + 7) & ~7;
        const size_t in_bytes = (size_t)(N / 4 + 2) * N * Wq * 16;
        float *in, *out;
        unsigned* cnt;
        CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, n * 4 + (1 << 20))); CK(hipMalloc(&cnt, (size_t)64 << 20));
        CK(hipMemset(in, 0, in_bytes)); CK(hipMemset(out, 0, n * 4));
        printf("N = %d\n", N);
        // stores only: depth of the chunk, then pacing
        for (int dch : {4, 8, 16, 32, 64, 128, 256}) run<16, 32, 256>(in, out, cnt, N, dch, 2, 0, 16);
        for (int dch : {64, 256}) for (int delta : {1, 2, 4}) run<16, 32, 256>(in, out, cnt, N, dch, 2, delta, 16);
        run<16, 32, 256>(in, out, cnt, N, 256, 2, 1, 4); run<16, 32, 256>(in, out, cnt, N, 256, 2, 1, 64);
        // stores only, occupancy limited to 4 workgroups per CU as in the cubic kernel (LDS padding)
        run<16, 32, 256>(in, out, cnt, N, 64, 2, 0, 16, 20480); run<16, 32, 256>(in, out, cnt, N, 64, 2, 1, 16, 20480);
        // loads only
        for (int dch : {16, 32, 64, 128, 256}) run<16, 32, 256>(in, out, cnt, N, dch, 1, 0, 16);
        // copy
        for (int dch : {16, 32, 64, 128, 256}) run<16, 32, 256>(in, out, cnt, N, dch, 3, 0, 16);
        for (int dch : {32, 64, 128, 256}) for (int delta : {1, 2, 4}) run<16, 32, 256>(in, out, cnt, N, dch, 3, delta, 16);
        run<16, 32, 256>(in, out, cnt, N, 128, 3, 2, 4); run<16, 32, 256>(in, out, cnt, N, 128, 3, 2, 64);
        for (int dch : {32, 128}) { run<32, 32, 256>(in, out, cnt, N, dch, 3, 0, 16); run<32, 32, 256>(in, out, cnt, N, dch, 3, 2, 16); }
        for (int dch : {32, 128}) { run<32, 32, 512>(in, out, cnt, N, dch, 3, 0, 16); run<32, 32, 512>(in, out, cnt, N, dch, 3, 2, 16); }
        // copy at 4 workgroups per CU
        for (int dch : {64, 256}) { run<16, 32, 256>(in, out, cnt, N, dch, 3, 0, 16, 20480); run<16, 32, 256>(in, out, cnt, N, dch, 3, 2, 16, 20480); }
        CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(cnt));
    }
    return 0;
}
