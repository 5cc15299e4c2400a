// Which side of the marching kernels' memory traffic is below the chip's streaming rate: the tile-shaped stores or the
// footprint-shaped loads?  No arithmetic at all; every kernel moves N^3 floats in ONE direction (the other direction is
// registers / LDS only), so the two sides can be priced separately and by geometry.
//   W  : write-only.  linear (1 KiB per wave-instruction) vs TH x TW tiles marching DCH planes, 4 B per lane
//        (row segments of 4*TW bytes), plain / nontemporal, output pitch N or N+8.
//   R  : read-only through LDS-DMA (buffer_load ... lds, 16 B per lane, two ring slots, barrier per step, data never
//        read back): rows of a TH x TW tile's footprint from the plane-quad layout ((TW+2) x 16 B runs, 4 planes per step)
//        or from the plain layout ((TW+4) x 4 B runs, one plane per step); linear reads for comparison.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/pattern_probe.hip -o gpurun_out/pattern_probe
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

template <int NTS>
__global__ __launch_bounds__(256) void w_linear(float4* __restrict__ out, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    if (i < n4) {
        if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(out + i));
        else *reinterpret_cast<v4f*>(out + i) = v;
    }
}

// TH x TW tile, 256 threads, 4 B per lane; marches dch planes; pitch = floats per output row
template <int TH, int TW, int NTS>
__global__ __launch_bounds__(256) void w_tiles(float* __restrict__ out, int N, int pitch, int dch, int nTh, int nTw)
{
    constexpr int RP = 256 / TW, NPIX = TH / RP;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tw_i = t % nTw, t2 = t / nTw, th_i = t2 % nTh, chunk = t2 / nTh;
    const int kw = threadIdx.x % TW, jh0 = threadIdx.x / TW;
    const size_t plane = (size_t)N * pitch;
    float* o = out + (size_t)chunk * dch * plane + (size_t)(th_i * TH + jh0) * pitch + tw_i * TW + kw;
    for (int d = 0; d < dch; ++d, o += plane) {
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
            const float v = (float)(d + px);
            if (NTS) __builtin_nontemporal_store(v, o + (size_t)px * RP * pitch);
            else o[(size_t)px * RP * pitch] = v;
        }
    }
}

// TH x TW tile, 16 B per lane (4 consecutive w per lane), 256 threads
template <int TH, int TW, int NTS>
__global__ __launch_bounds__(256) void w_tiles16(float* __restrict__ out, int N, int pitch, int dch, int nTh, int nTw)
{
    constexpr int TV = TW / 4, RP = 256 / TV, NPIX = (TH + RP - 1) / RP;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tw_i = t % nTw, t2 = t / nTw, th_i = t2 % nTh, chunk = t2 / nTh;
    const int kv = threadIdx.x % TV, jh0 = threadIdx.x / TV;
    const size_t plane = (size_t)N * pitch;
    float* o = out + (size_t)chunk * dch * plane + (size_t)(th_i * TH + jh0) * pitch + tw_i * TW + 4 * kv;
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    if (jh0 >= TH) return;
    for (int d = 0; d < dch; ++d, o += plane) {
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
            if (jh0 + px * RP < TH) {
                if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(o + (size_t)px * RP * pitch));
                else *reinterpret_cast<v4f*>(o + (size_t)px * RP * pitch) = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void r_linear(const float4* __restrict__ in, float* __restrict__ sink, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const float4 v = in[i];
        if (v.x == 12345.678f) sink[0] = v.y;
    }
}

// LDS-DMA reads of footprint rows, never read back.  QUAD = 1: layout [z/4][y][Wq][4], one step = one quad, rows of
// (TW + 2) positions x 16 B; QUAD = 0: plain [z][y][P], one step = one plane, rows of (TW + 4) floats (16-byte vectors).
template <int TH, int TW, int QUAD>
__global__ __launch_bounds__(256) void r_foot(const float* __restrict__ in, int N, int rowbytes, int steps_per_chunk, int nTh, int nTw, int halo)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 256;
    const int tid = threadIdx.x;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tw_i = t % nTw, t2 = t / nTw, th_i = t2 % nTh, chunk = t2 / nTh;
    const int rows = TH + 1 + 2 * halo;
    const int vec_per_row = QUAD ? (TW + 2 + 2 * halo) : (TW + 4 + 4 * halo) / 4;
    const int nvec = rows * vec_per_row;
    const int nvec64 = (nvec + 63) & ~63;
    const int slot_bytes = nvec64 * 16;
    int voff[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int v = tid + NT * it;
        if (v >= nvec) v = 0;
        const int y = v / vec_per_row, cx = v - y * vec_per_row;
        int gy = th_i * TH + y, gx = tw_i * TW * (QUAD ? 16 : 4) + cx * 16;       // bytes along the row
        if (gy >= N) gy = N - 1;
        if (gx + 16 > rowbytes) gx = 0;
        voff[it] = gy * rowbytes + gx;
    }
    const int nit = (nvec64 + NT - 1) / NT;
    const int step_bytes = N * rowbytes;                                // bytes of one quad-plane / plane
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(in) + (size_t)chunk * steps_per_chunk * step_bytes), 0, 0x7fffffff, 0x00020000);
    char* lds_c = reinterpret_cast<char*>(lds);
    auto issue = [&](int s, int slot_off) {
        char* dst = lds_c + slot_off + 16 * wave_first;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (it < nit && wave_first + NT * it < nvec64)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + 16 * NT * it), 16, voff[it], s * step_bytes, 0, 0);
    };
    int slot = 0;
    issue(0, 0);
    for (int s = 0; s < steps_per_chunk; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 1 < steps_per_chunk) issue(s + 1, slot ^ slot_bytes);
        slot ^= slot_bytes;
    }
}

template <typename F>
static float time_ms(F f, int iters)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    float r[5];
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&r[rep], a, b));
    }
    CK(hipGetLastError());
    for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (r[j] < r[i]) { float t = r[i]; r[i] = r[j]; r[j] = t; }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return r[2] / iters;
}

template <int TH, int TW, int NTS>
static void run_w(float* out, int N, int pitch, int dch)
{
    const int nTh = N / TH, nTw = N / TW, grid = nTh * nTw * (N / dch);
    float ms = time_ms([&] { hipLaunchKernelGGL((w_tiles<TH, TW, NTS>), dim3(grid), dim3(256), 0, 0, out, N, pitch, dch, nTh, nTw); }, 10);
    printf("  W tiles  %2dx%-3d  4B/lane nt=%d pitch=%4d dch=%3d : %.4f ms  %.2f TB/s\n", TH, TW, NTS, pitch, dch, ms, 4.0 * N * N * N / ms / 1e9);
}
template <int TH, int TW, int NTS>
static void run_w16(float* out, int N, int pitch, int dch)
{
    const int nTh = N / TH, nTw = N / TW, grid = nTh * nTw * (N / dch);
    float ms = time_ms([&] { hipLaunchKernelGGL((w_tiles16<TH, TW, NTS>), dim3(grid), dim3(256), 0, 0, out, N, pitch, dch, nTh, nTw); }, 10);
    printf("  W tiles  %2dx%-3d 16B/lane nt=%d pitch=%4d dch=%3d : %.4f ms  %.2f TB/s\n", TH, TW, NTS, pitch, dch, ms, 4.0 * N * N * N / ms / 1e9);
}

// TH x TW tile writer through a buffer descriptor: AUX = cache policy bits of the store (0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1, 18 sc1 nt);
// ORDER 0 = XCD-contiguous tile ids (w fastest, then h, then chunk), 1 = plain blockIdx, 2 = chunk fastest, 3 = XCD-contiguous with
// 4 planes per step written back to back (the quad kernel's store shape: a step = 4 planes x NPIX stores)
template <int TH, int TW, int AUX, int ORDER>
__global__ __launch_bounds__(256) void w_tiles_buf(float* __restrict__ out, int N, int dch, int nTh, int nTw)
{
    constexpr int RP = 256 / TW, NPIX = TH / RP;
    int t = blockIdx.x;
    if (ORDER != 1) t = xcd_contiguous(blockIdx.x, gridDim.x);
    int tw_i, th_i, chunk;
    if (ORDER == 2) { const int nch = gridDim.x / (nTh * nTw); chunk = t % nch; const int t2 = t / nch; tw_i = t2 % nTw; th_i = t2 / nTw; }
    else { tw_i = t % nTw; const int t2 = t / nTw; th_i = t2 % nTh; chunk = t2 / nTh; }
    const int kw = threadIdx.x % TW, jh0 = threadIdx.x / TW;
    const size_t plane = (size_t)N * N;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(out + (size_t)chunk * dch * plane + (size_t)(th_i * TH) * N + tw_i * TW), 0, 0x7fffffff, 0x00020000);
    int ob[NPIX];
#pragma unroll
    for (int px = 0; px < NPIX; ++px) ob[px] = ((jh0 + px * RP) * N + kw) * 4;
    const int pb = N * N * 4;
    for (int d = 0; d < dch; ++d) {
#pragma unroll
        for (int px = 0; px < NPIX; ++px)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(d + px)), rs, ob[px], d * pb, AUX);
    }
}
template <int TH, int TW, int AUX, int ORDER>
static void run_wb(float* out, int N, int dch)
{
    const int nTh = N / TH, nTw = N / TW, grid = nTh * nTw * (N / dch);
    float ms = time_ms([&] { hipLaunchKernelGGL((w_tiles_buf<TH, TW, AUX, ORDER>), dim3(grid), dim3(256), 0, 0, out, N, dch, nTh, nTw); }, 10);
    printf("  W buf    %2dx%-3d aux=%2d order=%d dch=%3d : %.4f ms  %.2f TB/s\n", TH, TW, AUX, ORDER, dch, ms, 4.0 * N * N * N / ms / 1e9);
}
template <int TH, int TW, int QUAD>
static void run_r(const float* in, int N, int halo, int dch)
{
    const int Wq = (N + 1 + 7) & ~7, P = ((N + 3) & ~3) + 4;
    const int rowbytes = QUAD ? Wq * 16 : P * 4;
    const int steps = QUAD ? dch / 4 : dch;
    const int nTh = N / TH, nTw = N / TW, grid = nTh * nTw * (N / dch);
    const int rows = TH + 1 + 2 * halo, vpr = QUAD ? (TW + 2 + 2 * halo) : (TW + 4 + 4 * halo) / 4;
    const int nvec64 = (rows * vpr + 63) & ~63;
    const int ldsb = 2 * nvec64 * 16;
    void (*fn)(const float*, int, int, int, int, int, int) = r_foot<TH, TW, QUAD>;
    CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float ms = time_ms([&] { hipLaunchKernelGGL(fn, dim3(grid), dim3(256), ldsb, 0, in, N, rowbytes, steps, nTh, nTw, halo); }, 10);
    const double staged = (double)grid * steps * rows * vpr * 16.0;
    printf("  R foot   %2dx%-3d %s halo=%d dch=%3d lds=%6d : %.4f ms  %.2f TB/s compulsory (%.2f TB/s staged, x%.2f)\n", TH, TW, QUAD ? "quad " : "plain", halo,
           dch, ldsb, ms, 4.0 * N * N * N / ms / 1e9, staged / ms / 1e9, staged / (4.0 * N * N * N));
}

int main()
{
    for (int N : {512, 1024}) {
        const size_t n = (size_t)N * N * N;
        const size_t cap = (size_t)N * N * (N + 16) * 4 + ((size_t)1 << 20);
        float *in, *out, *sink;
        CK(hipMalloc(&in, cap * 2)); CK(hipMalloc(&out, cap)); CK(hipMalloc(&sink, 256));
        CK(hipMemset(in, 0, cap * 2)); CK(hipMemset(out, 0, cap));
        printf("N = %d\n", N);
        const size_t n4 = n / 4;
        float ms = time_ms([&] { hipLaunchKernelGGL(w_linear<0>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (float4*)out, n4); }, 10);
        printf("  W linear 16B/lane nt=0 : %.4f ms  %.2f TB/s\n", ms, 4.0 * n / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(w_linear<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (float4*)out, n4); }, 10);
        printf("  W linear 16B/lane nt=1 : %.4f ms  %.2f TB/s\n", ms, 4.0 * n / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(r_linear, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float4*)in, sink, n4); }, 10);
        printf("  R linear 16B/lane      : %.4f ms  %.2f TB/s\n", ms, 4.0 * n / ms / 1e9);
        for (int dch : {64}) {
            run_w<16, 32, 0>(out, N, N, dch); run_w<16, 32, 1>(out, N, N, dch);
            run_w<16, 32, 1>(out, N, N + 8, dch);
            run_w<8, 64, 1>(out, N, N, dch); run_w<4, 128, 1>(out, N, N, dch); run_w<2, 256, 1>(out, N, N, dch);
            run_w<16, 64, 1>(out, N, N, dch); run_w<32, 32, 1>(out, N, N, dch);
            run_w16<16, 32, 1>(out, N, N, dch); run_w16<16, 64, 1>(out, N, N, dch); run_w16<8, 128, 1>(out, N, N, dch); run_w16<16, 32, 0>(out, N, N, dch);
            run_r<16, 32, 1>(in, N, 0, dch); run_r<16, 32, 1>(in, N, 1, dch);
            run_r<16, 32, 0>(in, N, 0, dch); run_r<16, 32, 0>(in, N, 1, dch);
            run_r<16, 64, 1>(in, N, 0, dch); run_r<32, 32, 1>(in, N, 0, dch); run_r<8, 32, 1>(in, N, 0, dch);
        }
        run_wb<16, 32, 0, 0>(out, N, 64); run_wb<16, 32, 1, 0>(out, N, 64); run_wb<16, 32, 2, 0>(out, N, 64); run_wb<16, 32, 16, 0>(out, N, 64);
        run_wb<16, 32, 17, 0>(out, N, 64); run_wb<16, 32, 18, 0>(out, N, 64);
        run_wb<16, 32, 0, 1>(out, N, 64); run_wb<16, 32, 0, 2>(out, N, 64); run_wb<16, 32, 2, 1>(out, N, 64); run_wb<16, 32, 2, 2>(out, N, 64);
        run_wb<16, 32, 0, 0>(out, N, 1); run_wb<16, 32, 0, 0>(out, N, 4); run_wb<16, 32, 0, 0>(out, N, 8); run_wb<16, 32, 0, 0>(out, N, 256);
        run_wb<16, 32, 2, 0>(out, N, 1); run_wb<16, 32, 2, 0>(out, N, 4); run_wb<16, 32, 0, 1>(out, N, 1); run_wb<16, 32, 0, 1>(out, N, 4);
        run_wb<32, 32, 0, 0>(out, N, 64); run_wb<32, 32, 0, 0>(out, N, 4); run_wb<16, 64, 0, 0>(out, N, 4);
        run_w<16, 32, 1>(out, N, N, 16); run_w<16, 32, 1>(out, N, N, 128);
        run_r<16, 32, 1>(in, N, 0, 16); run_r<16, 32, 1>(in, N, 0, 128);
        CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(sink));
    }
    return 0;
}
