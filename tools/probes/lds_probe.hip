// Does gfx950 serve ds_read_b128 / ds_read_b64 from addresses that are only 4-byte aligned, and at what rate?
// (the cubic general-rotation gather reads 4 consecutive floats per tap row at an arbitrary column)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/lds_probe.hip -o tools/probes/lds_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: 4 x ds_read_b32 (offsets 0,4,8,12)   MODE 1: ds_read_b128   MODE 2: 2 x ds_read_b64   MODE 3: 2 x ds_read2_b32
template <int MODE>
__global__ __launch_bounds__(256) void probe(const int* __restrict__ idx, float* __restrict__ out, int iters, int check)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds;
    float acc = 0.f;
    unsigned a = base + 4u * (unsigned)idx[threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        float r0, r1, r2, r3;
        if (MODE == 0) {
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a));
        } else if (MODE == 1) {
            v4f v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a));
            r0 = v.x; r1 = v.y; r2 = v.z; r3 = v.w;
        } else if (MODE == 2) {
            v2f u, w;
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(u), "=&v"(w) : "v"(a));
            r0 = u.x; r1 = u.y; r2 = w.x; r3 = w.y;
        } else {
            v2f u, w;
            asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(u), "=&v"(w) : "v"(a));
            r0 = u.x; r1 = u.y; r2 = w.x; r3 = w.y;
        }
        if (check && it == 0) {
            const int i0 = idx[threadIdx.x];
            out[threadIdx.x] = (r0 == (float)i0 && r1 == (float)(i0 + 1) && r2 == (float)(i0 + 2) && r3 == (float)(i0 + 3)) ? 1.f : 0.f;
        }
        acc += r0 + r1 + r2 + r3;
        a += (it & 1) ? 64u : (unsigned)-64;       // wander a little, keep alignment class
    }
    if (!check && acc == 12345.f) out[threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, const std::vector<int>& h_idx, int* d_idx, float* d_out)
{
    CK(hipMemcpy(d_idx, h_idx.data(), 256 * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256), 0, 0, d_idx, d_out, 1, 1);
    std::vector<float> ok(256);
    CK(hipMemcpy(ok.data(), d_out, 256 * sizeof(float), hipMemcpyDeviceToHost));
    int good = 0; for (float f : ok) good += f == 1.f;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 2000, blocks = 1024;
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_idx, d_out, iters, 0);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_idx, d_out, iters, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // clocks per wave-level "4 floats per lane" fetch on one CU: blocks/256 CUs sequentially x 4 waves share the LDS pipe
    const double fetches_per_cu = (double)blocks / 256.0 * 4.0 * iters;
    printf("  %-14s correct lanes %3d/256   %.3f ms  -> %.1f ns per wave fetch per CU (16 B x 64 lanes)\n", name, good, ms, ms * 1e6 / fetches_per_cu);
}

int main()
{
    int* d_idx; float* d_out;
    CK(hipMalloc(&d_idx, 256 * sizeof(int))); CK(hipMalloc(&d_out, 256 * sizeof(float)));
    struct Pat { const char* name; int stride; int off; bool rnd; };
    const Pat pats[] = {{"aligned, stride 4 floats", 4, 0, false}, {"offset 1, stride 4", 4, 1, false}, {"offset 2, stride 4", 4, 2, false},
                        {"offset 3, stride 4", 4, 3, false}, {"stride 5 (mixed alignment)", 5, 0, false}, {"stride 37", 37, 1, false},
                        {"rotated-gather-like (random rows)", 0, 0, true}};
    for (const Pat& p : pats) {
        std::vector<int> idx(256);
        srand(7);
        for (int t = 0; t < 256; ++t) {
            if (p.rnd) { const int l = t & 63; idx[t] = 2048 + ((int)(l * 0.6) * 44 * 28 % 8192) + (int)(l * 0.55) * 44 + (int)(l * 0.57) + (t >> 6) * 3; idx[t] %= 12000; }
            else idx[t] = 1024 + (t & 63) * p.stride + p.off + (t >> 6) * 2048;
        }
        printf("%s\n", p.name);
        run<0>("4 x b32", idx, d_idx, d_out);
        run<1>("b128", idx, d_idx, d_out);
        run<2>("2 x b64", idx, d_idx, d_out);
        run<3>("2 x read2_b32", idx, d_idx, d_out);
    }
    return 0;
}
