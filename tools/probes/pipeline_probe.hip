// One-shot pipeline rehearsal outside the library: pitched chunk uploads into ONE resident buffer, a kernel per output slab
// gated on the uploads it needs, chunk downloads gated on the kernels.  Which combination of stream flags / dependency
// mechanism keeps PCIe duplex (target: ~11.7 ms for 512 MiB up + 512 MiB down; sequential: ~19 ms)?
// build: hipcc -O2 --offload-arch=gfx950 tools/probes/pipeline_probe.hip -o /tmp/pipeline_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// out[d][h][w] = 0.5 * (src[d][h][w] + src[d+1][h][w])  (needs one plane of the next chunk: a halo dependency)
__global__ void slab_kernel(const float* __restrict__ src, float* __restrict__ out, int D, int H, int W, int P, int d0, int d1)
{
    const int64_t n = (int64_t)(d1 - d0) * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W); const int64_t t = i / W; const int h = (int)(t % H); const int d = d0 + (int)(t / H);
        const float a = src[((int64_t)d * H + h) * P + w];
        const float b = d + 1 < D ? src[((int64_t)(d + 1) * H + h) * P + w] : 0.f;
        out[((int64_t)d * H + h) * W + w] = 0.5f * (a + b);
    }
}

int main(int argc, char** argv)
{
    const int D = 512, H = 512, W = 512, P = 516, nch = 16, Dc = D / nch;
    const size_t N = (size_t)D * H * W * 4;
    float *h_in = (float*)aligned_alloc(4096, N), *h_out = (float*)aligned_alloc(4096, N);
    for (size_t i = 0; i < N / 4; ++i) h_in[i] = (float)(i % 1000);
    memset(h_out, 0, N);
    CK(hipHostRegister(h_in, N, hipHostRegisterDefault)); CK(hipHostRegister(h_out, N, hipHostRegisterDefault));
    float *d_src, *d_out;
    CK(hipMalloc((void**)&d_src, (size_t)D * H * P * 4)); CK(hipMalloc((void**)&d_out, N));
    CK(hipMemset(d_src, 0, (size_t)D * H * P * 4));
    for (int kflag = 0; kflag < 2; ++kflag) {            // kernel stream: blocking (hipStreamDefault) or non-blocking
        for (int mode = 0; mode < 3; ++mode) {           // 0 = stream waits, all enqueued at once; 1 = host waits; 2 = sequential reference
            hipStream_t s_up, s_dn, s_k;
            CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_dn, hipStreamNonBlocking));
            CK(hipStreamCreateWithFlags(&s_k, kflag ? hipStreamNonBlocking : hipStreamDefault));
            std::vector<hipEvent_t> ev_up(nch), ev_k(nch);
            for (int k = 0; k < nch; ++k) { CK(hipEventCreateWithFlags(&ev_up[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_k[k], hipEventDisableTiming)); }
            for (int rep = 0; rep < 3; ++rep) {
                const double t0 = now();
                auto up = [&](int k) {
                    CK(hipMemcpy2DAsync(d_src + (size_t)k * Dc * H * P, (size_t)P * 4, h_in + (size_t)k * Dc * H * W, (size_t)W * 4, (size_t)W * 4,
                                        (size_t)Dc * H, hipMemcpyHostToDevice, s_up));
                    CK(hipEventRecord(ev_up[k], s_up));
                };
                auto kern = [&](int j) { hipLaunchKernelGGL(slab_kernel, dim3(2048), dim3(256), 0, s_k, d_src, d_out, D, H, W, P, j * Dc, (j + 1) * Dc); CK(hipGetLastError()); CK(hipEventRecord(ev_k[j], s_k)); };
                auto dn = [&](int j) { CK(hipMemcpyAsync(h_out + (size_t)j * Dc * H * W, d_out + (size_t)j * Dc * H * W, (size_t)Dc * H * W * 4, hipMemcpyDeviceToHost, s_dn)); };
                if (mode == 0) {
                    for (int k = 0; k < nch; ++k) up(k);
                    for (int j = 0; j < nch; ++j) {
                        CK(hipStreamWaitEvent(s_k, ev_up[j + 1 < nch ? j + 1 : j], 0));
                        kern(j);
                        CK(hipStreamWaitEvent(s_dn, ev_k[j], 0));
                        dn(j);
                    }
                } else if (mode == 1) {
                    up(0); up(1); if (nch > 2) up(2);
                    for (int j = 0; j < nch; ++j) {
                        CK(hipEventSynchronize(ev_up[j + 1 < nch ? j + 1 : j]));
                        kern(j);
                        CK(hipEventSynchronize(ev_k[j]));
                        dn(j);
                        if (j + 3 < nch) up(j + 3);
                    }
                } else {
                    for (int k = 0; k < nch; ++k) up(k);
                    CK(hipStreamSynchronize(s_up));
                    for (int j = 0; j < nch; ++j) kern(j);
                    CK(hipStreamSynchronize(s_k));
                    for (int j = 0; j < nch; ++j) dn(j);
                }
                const double ti = now() - t0;
                CK(hipStreamSynchronize(s_up)); CK(hipStreamSynchronize(s_k)); CK(hipStreamSynchronize(s_dn));
                const double tt = now() - t0;
                // verify a few values
                bool ok = true;
                for (int d = 0; d < D; d += 37) {
                    const size_t i = ((size_t)d * H + 11) * W + 13;
                    const float want = 0.5f * (h_in[i] + (d + 1 < D ? h_in[i + (size_t)H * W] : 0.f));
                    if (h_out[i] != want) ok = false;
                }
                printf("kernel stream %s, %s: issue %.2f ms, total %.2f ms %s\n", kflag ? "non-blocking" : "blocking",
                       mode == 0 ? "stream waits, enqueue all" : mode == 1 ? "host waits, 3 uploads ahead" : "sequential", ti, tt, ok ? "" : "WRONG");
                memset(h_out, 0, 4096);
            }
            for (int k = 0; k < nch; ++k) { CK(hipEventDestroy(ev_up[k])); CK(hipEventDestroy(ev_k[k])); }
            CK(hipStreamDestroy(s_up)); CK(hipStreamDestroy(s_dn)); CK(hipStreamDestroy(s_k));
        }
    }
    return 0;
}
