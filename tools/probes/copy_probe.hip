// What does the memory system give a kernel that reads N^3 floats and writes N^3 floats?  (the ceiling of the
// transform's 8 B/voxel roofline on this box, for the access patterns the transform kernels use)
//   linear16 : out[i] = in[i], 16 B per lane, one pass
//   tiles4   : workgroup = TH x TW in-plane tile marching through DCH planes, 4 B per lane (row segments of TW*4 bytes),
//              XCD-contiguous tile order as in the marching kernels
//   tiles_lds: same, but the source goes global -> LDS (direct, 16 B per lane) -> registers -> global, with the
//              marching kernels' ring / barrier structure (G planes per barrier, LA groups ahead)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/copy_probe.hip -o gpurun_out/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int xcd_contiguous(int b, int n)
{
    const int xcd = b & 7, q = n >> 3, r = n & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (b >> 3);
}

__global__ __launch_bounds__(256) void linear16(const float4* __restrict__ in, float4* __restrict__ out, size_t n4)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = in[i];
}

template <int TH, int TW, int NT>
__global__ __launch_bounds__(NT) void tiles4(const float* __restrict__ in, float* __restrict__ out, int N, int dch, int nTh, int nTw)
{
    constexpr int RP = NT / TW, NPIX = TH / RP;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tw_i = t % nTw, t2 = t / nTw, th_i = t2 % nTh, chunk = t2 / nTh;
    const int kw = threadIdx.x % TW, jh0 = threadIdx.x / TW;
    const size_t plane = (size_t)N * N;
    size_t off[NPIX];
#pragma unroll
    for (int px = 0; px < NPIX; ++px) off[px] = (size_t)chunk * dch * plane + (size_t)(th_i * TH + jh0 + px * RP) * N + tw_i * TW + kw;
    for (int d = 0; d < dch; ++d) {
        float v[NPIX];
#pragma unroll
        for (int px = 0; px < NPIX; ++px) v[px] = in[off[px] + d * plane];
#pragma unroll
        for (int px = 0; px < NPIX; ++px) out[off[px] + d * plane] = v[px];
    }
}


// 16 B per lane for loads and stores; ORDER: 0 = xcd-contiguous tiles (w fastest), 1 = plain blockIdx, 2 = chunk fastest
// NTS: nontemporal stores
template <int TH, int TW, int ORDER, int NTS>
__global__ __launch_bounds__(256) void tiles16(const float* __restrict__ in, float* __restrict__ out, int N, int dch, int nTh, int nTw)
{
    constexpr int TV = TW / 4, RP = 256 / TV, NPIX = TH / RP;
    static_assert(TH % RP == 0 && NPIX >= 1, "mapping");
    int t = blockIdx.x;
    if (ORDER == 0) t = xcd_contiguous(blockIdx.x, gridDim.x);
    int tw_i, th_i, chunk;
    if (ORDER == 2) { const int nch = gridDim.x / (nTh * nTw); t = xcd_contiguous(blockIdx.x, gridDim.x); chunk = t % nch; const int t2 = t / nch; tw_i = t2 % nTw; th_i = t2 / nTw; }
    else { tw_i = t % nTw; const int t2 = t / nTw; th_i = t2 % nTh; chunk = t2 / nTh; }
    const int kv = threadIdx.x % TV, jh0 = threadIdx.x / TV;
    const size_t plane = (size_t)N * N;
    size_t off[NPIX];
#pragma unroll
    for (int px = 0; px < NPIX; ++px) off[px] = (size_t)chunk * dch * plane + (size_t)(th_i * TH + jh0 + px * RP) * N + tw_i * TW + 4 * kv;
    for (int d = 0; d < dch; ++d) {
        float4 v[NPIX];
#pragma unroll
        for (int px = 0; px < NPIX; ++px) v[px] = *reinterpret_cast<const float4*>(in + off[px] + d * plane);
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
            float4* o = reinterpret_cast<float4*>(out + off[px] + d * plane);
            if (NTS) { __builtin_nontemporal_store(v[px].x, &o->x); __builtin_nontemporal_store(v[px].y, &o->y); __builtin_nontemporal_store(v[px].z, &o->z); __builtin_nontemporal_store(v[px].w, &o->w); }
            else *o = v[px];
        }
    }
}

// 4 B per lane, ORDER / nontemporal variants of tiles4 (16x32 tile)
template <int ORDER, int NTS, int NTL>
__global__ __launch_bounds__(256) void tiles4v(const float* __restrict__ in, float* __restrict__ out, int N, int dch, int nTh, int nTw)
{
    constexpr int TH = 16, TW = 32, RP = 256 / TW, NPIX = TH / RP;
    int t = blockIdx.x;
    if (ORDER == 0) t = xcd_contiguous(blockIdx.x, gridDim.x);
    int tw_i, th_i, chunk;
    if (ORDER == 2) { const int nch = gridDim.x / (nTh * nTw); t = xcd_contiguous(blockIdx.x, gridDim.x); chunk = t % nch; const int t2 = t / nch; tw_i = t2 % nTw; th_i = t2 / nTw; }
    else { tw_i = t % nTw; const int t2 = t / nTw; th_i = t2 % nTh; chunk = t2 / nTh; }
    const int kw = threadIdx.x % TW, jh0 = threadIdx.x / TW;
    const size_t plane = (size_t)N * N;
    size_t off[NPIX];
#pragma unroll
    for (int px = 0; px < NPIX; ++px) off[px] = (size_t)chunk * dch * plane + (size_t)(th_i * TH + jh0 + px * RP) * N + tw_i * TW + kw;
    for (int d = 0; d < dch; ++d) {
        float v[NPIX];
#pragma unroll
        for (int px = 0; px < NPIX; ++px) v[px] = NTL ? __builtin_nontemporal_load(in + off[px] + d * plane) : in[off[px] + d * plane];
#pragma unroll
        for (int px = 0; px < NPIX; ++px) {
            if (NTS) __builtin_nontemporal_store(v[px], out + off[px] + d * plane);
            else out[off[px] + d * plane] = v[px];
        }
    }
}


// which side is sensitive to the tiled pattern?  MODE 0: tiled reads (16x64 tiles marching 16 planes), linear writes;
// MODE 1: linear reads, tiled writes.  (the copy is then a permutation, same bytes)
template <int MODE>
__global__ __launch_bounds__(256) void rw_mix(const float* __restrict__ in, float* __restrict__ out, int N, int dch, int nTh, int nTw)
{
    constexpr int TH = 16, TW = 64, TV = TW / 4, RP = 256 / TV;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tw_i = t % nTw, t2 = t / nTw, th_i = t2 % nTh, chunk = t2 / nTh;
    const int kv = threadIdx.x % TV, jh0 = threadIdx.x / TV;
    const size_t plane = (size_t)N * N;
    const size_t toff = (size_t)chunk * dch * plane + (size_t)(th_i * TH + jh0) * N + tw_i * TW + 4 * kv;
    const size_t loff = ((size_t)t * dch) * (TH * TW) + threadIdx.x * 4;     // this workgroup's contiguous dch*4 KB
    for (int d = 0; d < dch; ++d) {
        const float4 v = *reinterpret_cast<const float4*>(in + (MODE == 0 ? toff + d * plane : loff + (size_t)d * TH * TW));
        *reinterpret_cast<float4*>(out + (MODE == 0 ? loff + (size_t)d * TH * TW : toff + d * plane)) = v;
    }
}

// one full row (N floats = N/4 lanes x 16 B) per plane, marching dch planes: contiguous 4 KB pieces, plane jumps
__global__ __launch_bounds__(256) void rows16(const float* __restrict__ in, float* __restrict__ out, int N, int dch)
{
    const int rows_per_wg = 1024 / N;                       // N = 1024: 1 row; N = 512: 2 rows
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int nrow = N / rows_per_wg;
    const int r_i = t % nrow, chunk = t / nrow;
    const size_t plane = (size_t)N * N;
    const size_t off = (size_t)chunk * dch * plane + (size_t)r_i * rows_per_wg * N + threadIdx.x * 4;
    for (int d = 0; d < dch; ++d) {
        const float4 v = *reinterpret_cast<const float4*>(in + off + d * plane);
        *reinterpret_cast<float4*>(out + off + d * plane) = v;
    }
}

// linear copy, but consecutive workgroups are scattered over the whole array
__global__ __launch_bounds__(256) void linear16_scrambled(const float4* __restrict__ in, float4* __restrict__ out, unsigned nblocks_mask)
{
    const unsigned b = (blockIdx.x * 2654435761u) & nblocks_mask;
    const size_t i = (size_t)b * 256 + threadIdx.x;
    out[i] = in[i];
}
// linear copy, each workgroup copies `per` consecutive 4 KB pieces in a loop (persistent-ish)
__global__ __launch_bounds__(256) void linear16_loop(const float4* __restrict__ in, float4* __restrict__ out, int per)
{
    const size_t base = (size_t)xcd_contiguous(blockIdx.x, gridDim.x) * per * 256 + threadIdx.x;
    for (int k = 0; k < per; ++k) out[base + (size_t)k * 256] = in[base + (size_t)k * 256];
}

// LDS-staged: per plane the TH x TW tile (+ `extra` floats per row to mimic the footprint overfetch) goes through LDS
// (body in a __device__ function: a lambda that uses amdgcn builtins and is called from a __global__ template makes the
// host pass drop the kernel's stub silently)
template <int TH, int TW, int G, int LA>
__device__ __forceinline__ void tiles_lds_body(const float* __restrict__ in, float* __restrict__ out, int N, int dch, int nTh, int nTw)
{
    constexpr int NT = 256, RP = NT / TW, NPIX = TH / RP;
    constexpr int R = (LA + 1) * G;
    constexpr int SLOT = TH * TW;                 // floats
    constexpr int NV = SLOT / 4;                  // 16-byte vectors per plane
    constexpr int NIT = (NV + NT - 1) / NT;
    constexpr int WN0 = (LA - 1) * G * NIT + LA * G * NPIX;
    constexpr int WN = WN0 > 63 ? 63 : WN0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tw_i = t % nTw, t2 = t / nTw, th_i = t2 % nTh, chunk = t2 / nTh;
    const int kw = tid % TW, jh0 = tid / TW;
    const size_t plane = (size_t)N * N;
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int d_begin = chunk * dch;
    int voff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int v = tid + NT * it;
        const int y = v / (TW / 4), cx = v % (TW / 4);
        voff[it] = ((h0 + y) * N + w0 + 4 * cx) * 4;
    }
    size_t ooff[NPIX];
#pragma unroll
    for (int px = 0; px < NPIX; ++px) ooff[px] = (size_t)d_begin * plane + (size_t)(h0 + jh0 + px * RP) * N + w0 + kw;
    const int plane_bytes = N * N * 4;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(in) + (size_t)d_begin * plane_bytes), 0, 0x7fffffff, 0x00020000);
    const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
    auto issue = [&](int P, int slot) {
        float* dst = lds + slot * SLOT + 4 * wave_first;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (tid + NT * it < NV)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + 4 * NT * it), 16, voff[it], P * plane_bytes, 0, 0);
    };
    const int ngroups = dch / G;
    int P_next = 0, slot_next = 0;
    for (int g = 0; g < LA && g < ngroups; ++g)
        for (int c = 0; c < G; ++c) { issue(P_next++, slot_next); slot_next = (slot_next + 1 == R) ? 0 : slot_next + 1; }
    int slot_cur = 0;
    for (int g = 0; g < ngroups; ++g) {
        // outstanding after loads(g): loads(g+1..g+LA-1) and stores; simplest: count loads+stores issued after loads(g)
        if (g < LA || g + LA > ngroups) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WN) : "memory");
        __builtin_amdgcn_s_barrier();
        if (g + LA < ngroups)
            for (int c = 0; c < G; ++c) { issue(P_next++, slot_next); slot_next = (slot_next + 1 == R) ? 0 : slot_next + 1; }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const float* pl = lds + ((slot_cur + i) % R) * SLOT;
#pragma unroll
            for (int px = 0; px < NPIX; ++px) out[ooff[px] + (size_t)(g * G + i) * plane] = pl[(jh0 + px * RP) * TW + kw];
        }
        slot_cur = (slot_cur + G) % R;
    }
}

template <int TH, int TW, int G, int LA>
__global__ __launch_bounds__(256) void tiles_lds(const float* __restrict__ in, float* __restrict__ out, int N, int dch, int nTh, int nTw)
{
    tiles_lds_body<TH, TW, G, LA>(in, out, N, dch, nTh, nTw);
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
    return r[2] / iters;                                   // median of 5 repeats
}

template <int TH, int TW, int G, int LA>
static void run_lds(const float* in, float* out, int N, int dch)
{
    const int nTh = N / TH, nTw = N / TW, grid = nTh * nTw * (N / dch);
    const int ldsb = (LA + 1) * G * TH * TW * 4;
    void (*fn)(const float*, float*, int, int, int, int) = tiles_lds<TH, TW, G, LA>;
    CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float ms = time_ms([&] { hipLaunchKernelGGL(fn, dim3(grid), dim3(256), ldsb, 0, in, out, N, dch, nTh, nTw); }, 10);
    printf("  tiles_lds %2dx%-3d G=%d LA=%d dch=%3d lds=%6d : %.4f ms  %.0f GB/s\n", TH, TW, G, LA, dch, ldsb, ms, 8.0 * N * N * N / ms / 1e6);
}

template <int TH, int TW, int NT>
static void run_t4(const float* in, float* out, int N, int dch)
{
    const int nTh = N / TH, nTw = N / TW, grid = nTh * nTw * (N / dch);
    void (*fn)(const float*, float*, int, int, int, int) = tiles4<TH, TW, NT>;
    float ms = time_ms([&] { hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), 0, 0, in, out, N, dch, nTh, nTw); }, 10);
    printf("  tiles4    %2dx%-3d NT=%d dch=%3d : %.4f ms  %.0f GB/s\n", TH, TW, NT, dch, ms, 8.0 * N * N * N / ms / 1e6);
}

template <int TH, int TW, int ORDER, int NTS>
static void run_t16(const float* in, float* out, int N, int dch)
{
    const int nTh = N / TH, nTw = N / TW, grid = nTh * nTw * (N / dch);
    void (*fn)(const float*, float*, int, int, int, int) = tiles16<TH, TW, ORDER, NTS>;
    float ms = time_ms([&] { hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, in, out, N, dch, nTh, nTw); }, 10);
    printf("  tiles16   %2dx%-3d order=%d nts=%d dch=%3d : %.4f ms  %.0f GB/s\n", TH, TW, ORDER, NTS, dch, ms, 8.0 * N * N * N / ms / 1e6);
}
template <int ORDER, int NTS, int NTL>
static void run_t4v(const float* in, float* out, int N, int dch)
{
    const int nTh = N / 16, nTw = N / 32, grid = nTh * nTw * (N / dch);
    void (*fn)(const float*, float*, int, int, int, int) = tiles4v<ORDER, NTS, NTL>;
    float ms = time_ms([&] { hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, in, out, N, dch, nTh, nTw); }, 10);
    printf("  tiles4v   16x32  order=%d nts=%d ntl=%d dch=%3d : %.4f ms  %.0f GB/s\n", ORDER, NTS, NTL, dch, ms, 8.0 * N * N * N / ms / 1e6);
}

int main()
{
    for (int N : {512, 1024}) {
        const size_t n = (size_t)N * N * N;
        float *in, *out;
        CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
        CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 4));
        printf("N = %d\n", N);
        {
            const size_t n4 = n / 4;
            float ms = time_ms([&] { hipLaunchKernelGGL(linear16, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float4*)in, (float4*)out, n4); }, 10);
            printf("  linear16: %.4f ms  %.0f GB/s\n", ms, 8.0 * n / ms / 1e6);
            float ms2 = time_ms([&] { (void)hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0); }, 10);
            printf("  hipMemcpy D2D: %.4f ms  %.0f GB/s\n", ms2, 8.0 * n / ms2 / 1e6);
        }
        run_lds<16, 32, 2, 1>(in, out, N, 16);
        run_lds<16, 32, 4, 1>(in, out, N, 16);
        // one-shot workgroups: load G planes, one barrier, store, exit
        run_lds<16, 32, 2, 1>(in, out, N, 2);
        run_lds<16, 32, 4, 1>(in, out, N, 4);
        run_lds<16, 32, 8, 1>(in, out, N, 8);
        run_lds<16, 32, 16, 1>(in, out, N, 16);
        run_lds<16, 64, 4, 1>(in, out, N, 4);
        run_lds<16, 64, 8, 1>(in, out, N, 8);
        run_lds<32, 32, 8, 1>(in, out, N, 8);
        run_lds<8, 32, 8, 1>(in, out, N, 8);
        run_lds<8, 32, 16, 1>(in, out, N, 16);
        run_lds<16, 32, 2, 2>(in, out, N, 16);
        run_lds<16, 32, 8, 1>(in, out, N, 16);
        run_lds<16, 64, 2, 1>(in, out, N, 16);
        run_lds<16, 64, 16, 1>(in, out, N, 16);
        run_lds<32, 32, 16, 1>(in, out, N, 16);
        run_lds<32, 32, 4, 1>(in, out, N, 4);
        run_lds<32, 64, 4, 1>(in, out, N, 4);
        run_lds<32, 64, 8, 1>(in, out, N, 8);
        CK(hipFree(in)); CK(hipFree(out));
    }
    return 0;
}
