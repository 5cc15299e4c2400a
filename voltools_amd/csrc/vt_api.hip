// vt_api.hip -- the C ABI of include/voltools_hip.h (hipMalloc / hipMemcpyAsync / kernel launches).
#include "vt_internal.h"
#include "vt_device.h"
#include "vt_host.h"

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include <unistd.h>

using namespace vt;

// ---- VT_DEBUG_GUARD=1: every device allocation of this library gets 1 MiB of NaN (0xFF bytes) in front of it and behind
// it.  A kernel that reads out of bounds then produces NaN (parity tests fail instead of passing on whatever the neighbouring
// allocation held, or faulting when the neighbour has been freed); one that writes out of bounds is caught when the buffer
// is released (message + abort).  A poor man's address sanitizer for the GPU side (the real one is not available here). ----
namespace vt_guard {
constexpr size_t kGuard = (size_t)1 << 20;
struct Entry { void* base; size_t bytes; };
static std::mutex mu;
static std::unordered_map<void*, Entry> live;
static bool enabled()
{
    static const bool on = std::getenv("VT_DEBUG_GUARD") != nullptr;
    return on;
}
static hipError_t alloc(void** p, size_t bytes)
{
    if (!enabled()) return hipMalloc(p, bytes);
    char* base = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), bytes + 2 * kGuard);
    if (e != hipSuccess) return e;
    (void)hipMemset(base, 0xFF, kGuard);
    (void)hipMemset(base + kGuard + bytes, 0xFF, kGuard);
    (void)hipDeviceSynchronize();
    *p = base + kGuard;
    std::lock_guard<std::mutex> lk(mu);
    live[*p] = Entry{base, bytes};
    return hipSuccess;
}
static hipError_t release(void* p)
{
    if (!enabled() || !p) return hipFree(p);
    Entry en{nullptr, 0};
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) return hipFree(p);
        en = it->second;
        live.erase(it);
    }
    (void)hipDeviceSynchronize();
    std::vector<unsigned char> h(2 * kGuard);
    (void)hipMemcpy(h.data(), en.base, kGuard, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h.data() + kGuard, static_cast<char*>(en.base) + kGuard + en.bytes, kGuard, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < 2 * kGuard; ++i)
        if (h[i] != 0xFF) {
            std::fprintf(stderr, "[vt guard] out-of-bounds WRITE %s a device buffer of %zu bytes (guard offset %zu)\n",
                         i < kGuard ? "in front of" : "behind", en.bytes, i < kGuard ? kGuard - i : i - kGuard);
            std::fflush(stderr);
            std::abort();
        }
    return hipFree(en.base);
}
}  // namespace vt_guard
#define hipMalloc(p, n) vt_guard::alloc(reinterpret_cast<void**>(p), (n))
#define hipFree(p) vt_guard::release(p)

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define VT_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (void)hipGetLastError();   /* the runtime's last-error is sticky: a failure reported here must not resurface at the next launch check */ \
            return fail((int)e_, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                          \
    } while (0)

int use_device(int dev)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(VT_ENODEV, "no HIP device available (%s)", hipGetErrorString(e));
    if (dev < 0 || dev >= n) return fail(VT_ENODEV, "device index %d out of range (0..%d)", dev, n - 1);
    VT_HIP(hipSetDevice(dev));
    return 0;
}

std::once_flag g_init_flag[64];
hipError_t g_init_err[64];
// per-device constants, fetched once (hipGetDeviceProperties and a hipMalloc per handle were a visible part of the
// ~0.4 ms a one-shot transform of a tiny volume costs)
int g_cu_count[64], g_lds_limit[64];
float* g_zeros[64];                // 256 bytes of zeros per device: the border fetch target of the tiled kernels (never freed)


// ---------------------------------------------------------------------------------------------------
// Per-device recycling of what a handle's life costs besides the copies: a HIP stream, two events and small device buffers.
// A one-shot transform() of a tiny volume spent 0.38 of its 0.43 ms creating and destroying these
// (round-1 measurement).  Streams and events go back to a free list when a handle is destroyed (it synchronises
// first); device buffers up to 64 MiB are kept by exact size, at most 16 of them and 256 MiB in total per device.
// ---------------------------------------------------------------------------------------------------
struct DeviceCache {
    std::mutex mu;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events;
    std::vector<std::pair<size_t, void*>> bufs;
    size_t buf_bytes = 0;
    std::vector<std::pair<size_t, void*>> big;     // large buffers, most recently released last
    size_t big_bytes = 0;
    // one-shot pipeline: two NON-blocking copy streams (uploads / downloads), one pipelined call at a time per device
    std::mutex pipe_mu;
    hipStream_t copy_up = nullptr, copy_dn = nullptr, pipe_k = nullptr;
};
DeviceCache g_cache[64];
constexpr size_t kCacheBufMax = (size_t)64 << 20, kCacheTotalMax = (size_t)256 << 20;
// Large buffers (resident sources, result staging) are recycled too, a few of them: a transfer to or from a device
// allocation only reaches full PCIe duplex from its third pass on ([measured, round 1] 512 MiB up + down
// through a buffer from hipMalloc 19.4, 18.0, then 11.7 ms; a buffer allocated per call never gets there), so a one-shot
// transform() that allocated its buffers per call could not overlap its upload with its download.  With 288 GB of HBM the
// cache may hold 16 GiB; vt_device_trim releases it.
constexpr size_t kBigBufMax = (size_t)8 << 30, kBigTotalMax = (size_t)16 << 30;
constexpr size_t kBigCount = 6;

hipError_t cached_stream(int dev, hipStream_t* s)
{
    if (dev < 64) {
        std::lock_guard<std::mutex> lk(g_cache[dev].mu);
        if (!g_cache[dev].streams.empty()) { *s = g_cache[dev].streams.back(); g_cache[dev].streams.pop_back(); return hipSuccess; }
    }
    // a blocking stream: ordered against the legacy null stream (torch's default), like the reference, which does
    // everything on cupy's null stream (transforms.py:168, volume.py:66)
    return hipStreamCreateWithFlags(s, hipStreamDefault);
}
void recycle_stream(int dev, hipStream_t s)
{
    if (dev < 64) {
        std::lock_guard<std::mutex> lk(g_cache[dev].mu);
        if (g_cache[dev].streams.size() < 8) { g_cache[dev].streams.push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}
hipError_t cached_event(int dev, hipEvent_t* e)
{
    if (dev < 64) {
        std::lock_guard<std::mutex> lk(g_cache[dev].mu);
        if (!g_cache[dev].events.empty()) { *e = g_cache[dev].events.back(); g_cache[dev].events.pop_back(); return hipSuccess; }
    }
    return hipEventCreate(e);
}
void recycle_event(int dev, hipEvent_t e)
{
    if (dev < 64) {
        std::lock_guard<std::mutex> lk(g_cache[dev].mu);
        if (g_cache[dev].events.size() < 16) { g_cache[dev].events.push_back(e); return; }
    }
    (void)hipEventDestroy(e);
}
hipError_t cached_malloc(int dev, void** p, size_t bytes)
{
    if (dev < 64 && bytes <= kCacheBufMax) {
        std::lock_guard<std::mutex> lk(g_cache[dev].mu);
        auto& b = g_cache[dev].bufs;
        for (size_t i = 0; i < b.size(); ++i)
            if (b[i].first == bytes) {
                *p = b[i].second;
                g_cache[dev].buf_bytes -= bytes;
                b.erase(b.begin() + (long)i);
                return hipSuccess;
            }
    }
    if (dev < 64 && bytes > kCacheBufMax && bytes <= kBigBufMax) {
        std::lock_guard<std::mutex> lk(g_cache[dev].mu);
        auto& b = g_cache[dev].big;
        for (size_t i = b.size(); i-- > 0;)
            if (b[i].first == bytes) {
                *p = b[i].second;
                g_cache[dev].big_bytes -= bytes;
                b.erase(b.begin() + (long)i);
                return hipSuccess;
            }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && dev < 64) {
        // out of memory with buffers parked in the cache: release them and try once more
        (void)hipGetLastError();
        std::vector<std::pair<size_t, void*>> drop;
        {
            std::lock_guard<std::mutex> lk(g_cache[dev].mu);
            drop.swap(g_cache[dev].big);
            g_cache[dev].big_bytes = 0;
        }
        if (!drop.empty()) {
            for (auto& d : drop) (void)hipFree(d.second);
            e = hipMalloc(p, bytes);
        }
    }
    return e;
}
void cached_free(int dev, void* p, size_t bytes)
{
    if (!p) return;
    if (dev < 64 && bytes > kCacheBufMax && bytes <= kBigBufMax) {
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> lk(g_cache[dev].mu);
            auto& b = g_cache[dev].big;
            b.emplace_back(bytes, p);
            g_cache[dev].big_bytes += bytes;
            while (!b.empty() && (b.size() > kBigCount || g_cache[dev].big_bytes > kBigTotalMax)) {   // oldest first
                drop.push_back(b.front().second);
                g_cache[dev].big_bytes -= b.front().first;
                b.erase(b.begin());
            }
        }
        for (void* d : drop) (void)hipFree(d);
        return;
    }
    if (dev < 64 && bytes <= kCacheBufMax) {
        // most recently released last; when the cache is full the oldest entries make room (a cache that only ever kept its
        // first 16 buffers sent every later size to hipMalloc / hipFree: 0.24 -> 0.7 ms per one-shot call at 100^3 after a
        // run over other sizes)
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> lk(g_cache[dev].mu);
            auto& b = g_cache[dev].bufs;
            b.emplace_back(bytes, p);
            g_cache[dev].buf_bytes += bytes;
            while (!b.empty() && (b.size() > 16 || g_cache[dev].buf_bytes > kCacheTotalMax)) {
                drop.push_back(b.front().second);
                g_cache[dev].buf_bytes -= b.front().first;
                b.erase(b.begin());
            }
        }
        for (void* d : drop) (void)hipFree(d);
        return;
    }
    (void)hipFree(p);
}

int init_device(int dev)
{
    int rc = use_device(dev);
    if (rc) return rc;
    if (dev < 64) {
        std::call_once(g_init_flag[dev], [dev]() {
            g_init_err[dev] = init_affine_kernels();
            if (g_init_err[dev] == hipSuccess) g_init_err[dev] = init_quad_kernels();
            if (g_init_err[dev] == hipSuccess) g_init_err[dev] = init_block_kernels();
            if (g_init_err[dev] == hipSuccess) g_init_err[dev] = init_rows_kernels();
            if (g_init_err[dev] == hipSuccess) g_init_err[dev] = init_span_kernels();
            hipDeviceProp_t prop;
            if (g_init_err[dev] == hipSuccess) g_init_err[dev] = hipGetDeviceProperties(&prop, dev);
            if (g_init_err[dev] == hipSuccess) {
                g_cu_count[dev] = prop.multiProcessorCount;
                g_lds_limit[dev] = (int)std::min<size_t>(160 * 1024, prop.sharedMemPerBlock > 0 ? prop.sharedMemPerBlock : 65536);
                g_init_err[dev] = hipMalloc(reinterpret_cast<void**>(&g_zeros[dev]), 256);
            }
            if (g_init_err[dev] == hipSuccess) g_init_err[dev] = hipMemset(g_zeros[dev], 0, 256);
        });
        if (g_init_err[dev] != hipSuccess)
            return fail((int)g_init_err[dev], "kernel attribute setup failed: %s", hipGetErrorString(g_init_err[dev]));
    }
    return 0;
}

}  // namespace

namespace {

// Host <-> device copies of caller-owned (pageable) numpy buffers.  Pageable copies run at ~10 GB/s through the
// runtime's staging path; pinning the caller's buffer in place for the duration of the copy lets the DMA engines run at
// PCIe rate.  Falls back to the plain copy when registration is refused (VT_NO_PIN=1 disables it).
//
// What must never happen (round 5, the cause of the `Memory access fault by GPU` on a host address that rounds 1, 2 and 4 each met and
// explained differently; found with VT_DEBUG_PIN=1, profiles/r05_pin_trace.txt): a registration of ours that OVERLAPS a pin of the
// runtime's own.  The runtime serves a pageable transfer of more than 1 MiB by pinning the caller's pages in place, and keeps the last few
// such pins (a cache keyed by address and size, released first-in first-out or when the queue dies), long after the transfer and after the
// caller's array has been freed.  When the heap hands the same addresses to the next, larger array and that array's interior is
// registered here, two pinned objects cover the same pages: the registration succeeds, the copy resolves its host address to the OLD,
// smaller object and runs off its end -- or the old pin is dropped from the cache under the copy -- and the GPU faults inside a range that
// was registered "ok" a moment ago.  (The trace: four 1.3 MB results at 0x..3d8eef90 .. 0x..3dcbd840, then a 32 MB input at 0x..3d8eef90,
// interior [0x..3da00000, 0x..3f600000) registered ok, fault at 0x..3f3c7000.  What the trace does not show is the step in between: the
// small arrays were FREED, the heap gave their pages back, and the next array got new pages at the old addresses -- a replay of the same
// transfers inside one arena whose pages never go away does not fault, tools/diag/pin_probe_check.py: the stale object is a pin whose pages
// are gone.)  Two rules follow:
//   1. no transfer issued here is ever pinned by the runtime: every pageable piece goes in slices of kSlice = 512 KiB, half the runtime's
//      threshold, i.e. through its staging buffers (1-8 MiB arrays lose the pinned rate they used to get: 0.3 ms on 4 MiB);
//   2. nothing is registered over a range in which the runtime already knows a pinned object (somebody else's pageable transfer, a caller's
//      own registration): the range is probed every kSlice bytes -- a runtime pin is longer than 1 MiB, so none can hide between two probes.
struct PinnedScope {
    void* ptr = nullptr;
    bool pinned = false;
    static bool trace() { return std::getenv("VT_DEBUG_PIN") != nullptr; }      // (read per scope: a test switches it on for one call)
    // VT_DEBUG_PIN=1: to stderr; VT_DEBUG_PIN=<path with a slash>: appended to that file line by line (a trace that survives the abort of a
    // GPU fault under a test runner that captures stderr)
    static void say(const char* fmt, ...) __attribute__((format(printf, 1, 2)))
    {
        const char* e = std::getenv("VT_DEBUG_PIN");
        if (!e) return;
        FILE* f = std::strchr(e, '/') ? std::fopen(e, "a") : stderr;
        if (!f) return;
        va_list ap;
        va_start(ap, fmt);
        std::vfprintf(f, fmt, ap);
        va_end(ap);
        if (f != stderr) std::fclose(f); else std::fflush(f);
    }
    // The process heap [start of the [heap] mapping, current break): arrays that live there are never registered (rule 3 below).
    static bool in_brk_heap(const void* p, size_t bytes)
    {
        static std::atomic<uintptr_t> heap_lo{0};
        uintptr_t lo_ = heap_lo.load(std::memory_order_relaxed);
        if (!lo_) {
            if (FILE* f = std::fopen("/proc/self/maps", "r")) {
                char line[512];
                while (std::fgets(line, sizeof line, f))
                    if (std::strstr(line, "[heap]")) { lo_ = (uintptr_t)std::strtoull(line, nullptr, 16); break; }
                std::fclose(f);
            }
            if (lo_) heap_lo.store(lo_, std::memory_order_relaxed);
        }
        if (!lo_) return false;
        const uintptr_t a = reinterpret_cast<uintptr_t>(p), brk_now = reinterpret_cast<uintptr_t>(sbrk(0));
        return a + bytes > lo_ && a < brk_now;
    }
    // Memory the runtime already knows as pinned host memory (the Python layer's pooled result buffers, a caller's own
    // hipHostMalloc / hipHostRegister, the runtime's own pins): registering the same range a second time SUCCEEDS, and the matching
    // unregister at the end of the scope then strips the owner's registration ([measured, round 2] the pool's later
    // hipHostUnregister failed with "pointer does not correspond to a registered memory region", and the sticky error
    // failed the next kernel-launch check).
    static bool already_registered(const void* p)
    {
        hipPointerAttribute_t attr;
        const hipError_t e = hipPointerGetAttributes(&attr, p);
        if (e != hipSuccess) { (void)hipGetLastError(); return false; }
        return attr.type == hipMemoryTypeHost;
    }
    static bool any_registered(uintptr_t a, uintptr_t b)
    {
        for (uintptr_t x = a; x < b; x += kSlice)
            if (already_registered(reinterpret_cast<const void*>(x))) return true;
        return b > a && already_registered(reinterpret_cast<const void*>(b - 1));
    }
    // Only WHOLE 2 MiB units that lie INSIDE the caller's array are registered, and the copies below are cut at the ends of that
    // interior: a range rounded outward would pin memory that is not the caller's (its heap neighbours'), and the host's transparent huge
    // pages make 2 MiB the unit in which a pin can come and go without touching anything else.  The ragged head and tail of the array
    // (< 2 MiB each) travel as pageable slices.
    char* lo = nullptr;               // the registered interior [lo, hi) of the caller's range
    char* hi = nullptr;
    mutable hipStream_t last_stream = nullptr;
    mutable bool used_stream = false;
    static constexpr uintptr_t kUnit = 2u << 20;
    static constexpr size_t kSlice = 512u << 10;
    static constexpr size_t kHeapMax = 32u << 20;          // glibc's DEFAULT_MMAP_THRESHOLD_MAX on 64-bit
    PinnedScope(const void* p, size_t bytes)
    {
        static const bool off = std::getenv("VT_NO_PIN") != nullptr;
        if (trace() && p) say("[vt pin] scope %p + %zu%s\n", p, bytes, in_brk_heap(p, bytes) ? " (process heap)" : "");
        // Rule 3: nothing in the process heap is registered, and nothing the heap COULD hold: glibc serves requests up to 32 MiB from the
        // heap once its dynamic mmap threshold has risen (it rises to the size of every freed mapping up to that maximum), larger ones always
        // from mappings of their own.  Every fault seen -- rounds 1, 2, 4 and three this round -- was at a heap address, inside an interior
        // registered "ok" a moment before, after the heap had taken pages back and handed the addresses out again (profiles/r05_pin_trace.txt).
        if (off || !p || bytes <= kHeapMax || in_brk_heap(p, bytes)) return;
        const uintptr_t a0 = (reinterpret_cast<uintptr_t>(p) + kUnit - 1) & ~(kUnit - 1);
        const uintptr_t a1 = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(kUnit - 1);
        if (a1 <= a0 || a1 - a0 < (4u << 20)) return;
        // (the head and tail are probed too: a runtime pin that starts in the head reaches into the interior)
        if (any_registered(reinterpret_cast<uintptr_t>(p), reinterpret_cast<uintptr_t>(p) + bytes)) {
            if (trace()) say("[vt pin] %p + %zu overlaps memory the runtime has pinned already: not registered\n", p, bytes);
            return;
        }
        ptr = reinterpret_cast<void*>(a0);
        pinned = hipHostRegister(ptr, a1 - a0, hipHostRegisterDefault) == hipSuccess;
        if (trace()) say("[vt pin] registered [%p, %p): %s\n", ptr, (void*)a1, pinned ? "ok" : "refused");
        if (!pinned) { (void)hipGetLastError(); return; }
        lo = reinterpret_cast<char*>(a0);
        hi = reinterpret_cast<char*>(a1);
    }
    // one pageable piece, in slices the runtime stages (rule 1)
    static hipError_t sliced(char* dst, const char* src, size_t bytes, hipMemcpyKind kind, hipStream_t st)
    {
        // VT_PIN_UNSLICED=1 (diagnosis only: tools/diag/pin_probe_check.py): one transfer, which the runtime pins in place when it is large
        if (std::getenv("VT_PIN_UNSLICED")) return hipMemcpyAsync(dst, src, bytes, kind, st);
        for (size_t off = 0; off < bytes; off += kSlice) {
            const hipError_t e = hipMemcpyAsync(dst + off, src + off, std::min(kSlice, bytes - off), kind, st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    // A copy between device memory and a part of the caller's range, cut so that every piece lies either wholly inside the registered
    // interior or wholly outside it: the runtime decides "pinned or pageable" from a transfer's first host byte, and a pinned transfer
    // that runs past the end of the registration would let the DMA engine read unmapped pages.  `host` may be the source or the target.
    hipError_t copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) const
    {
        if (bytes == 0) return hipSuccess;
        last_stream = st; used_stream = true;
        const bool h2d = kind == hipMemcpyHostToDevice;
        const char* const h0 = h2d ? static_cast<const char*>(src) : static_cast<const char*>(dst);
        const char* const h1 = h0 + bytes;
        if (!pinned) {
            // memory that is registered as a whole (the Python layer's pooled result buffers, a caller's own pinned array): one transfer
            if (already_registered(h0) && already_registered(h1 - 1)) return hipMemcpyAsync(dst, src, bytes, kind, st);
            return sliced(static_cast<char*>(dst), static_cast<const char*>(src), bytes, kind, st);
        }
        const char* cuts[4] = {h0, std::min(std::max(h0, (const char*)lo), h1), std::min(std::max(h0, (const char*)hi), h1), h1};
        for (int i = 0; i < 3; ++i) {
            const size_t off = (size_t)(cuts[i] - h0), len = (size_t)(cuts[i + 1] - cuts[i]);
            if (!len) continue;
            const hipError_t e = (i == 1) ? hipMemcpyAsync(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len, kind, st)
                                          : sliced(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len, kind, st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    // ... and the pitched form: `rows` dense host rows of `row_bytes` to / from device rows `dpitch` apart.  Whole rows inside the interior
    // go as one 2-D copy, rows outside it in blocks of at most kSlice bytes, a row that straddles one of its ends is cut there.
    hipError_t copy2d(void* dev, size_t dpitch, const void* host, size_t row_bytes, size_t rows, bool h2d, hipStream_t st) const
    {
        const hipMemcpyKind kind = h2d ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
        auto block = [&](size_t r0, size_t r1) -> hipError_t {
            if (r1 <= r0) return hipSuccess;
            char* d = static_cast<char*>(dev) + r0 * dpitch;
            const char* h = static_cast<const char*>(host) + r0 * row_bytes;
            return h2d ? hipMemcpy2DAsync(d, dpitch, h, row_bytes, row_bytes, r1 - r0, kind, st)
                       : hipMemcpy2DAsync(const_cast<char*>(h), row_bytes, d, dpitch, row_bytes, r1 - r0, kind, st);
        };
        auto one_row = [&](size_t r) -> hipError_t {          // through copy(): cut at the interior's ends, sliced where pageable
            char* d = static_cast<char*>(dev) + r * dpitch;
            const char* h = static_cast<const char*>(host) + r * row_bytes;
            return h2d ? copy(d, h, row_bytes, kind, st) : copy(const_cast<char*>(h), d, row_bytes, kind, st);
        };
        auto pageable = [&](size_t r0, size_t r1) -> hipError_t {       // rows outside the interior: blocks the runtime stages (rule 1)
            if (r1 <= r0) return hipSuccess;
            if (row_bytes > kSlice) {
                for (size_t r = r0; r < r1; ++r) { const hipError_t e = one_row(r); if (e != hipSuccess) return e; }
                return hipSuccess;
            }
            const size_t step = std::max<size_t>(1, kSlice / row_bytes);
            for (size_t r = r0; r < r1; r += step) { const hipError_t e = block(r, std::min(r + step, r1)); if (e != hipSuccess) return e; }
            return hipSuccess;
        };
        if (rows == 0 || row_bytes == 0) return hipSuccess;
        last_stream = st; used_stream = true;
        const char* const h0 = static_cast<const char*>(host);
        if (!pinned) {
            if (already_registered(h0) && already_registered(h0 + rows * row_bytes - 1)) return block(0, rows);
            return pageable(0, rows);
        }
        auto row_of = [&](const char* a) -> size_t {          // row that holds byte a
            if (a <= h0) return 0;
            return std::min(rows, (size_t)(a - h0) / row_bytes);
        };
        // rows [0, ra): wholly below lo; row ra may straddle lo; rows (ra', rb): wholly inside; row rb may straddle hi; rows beyond: outside
        const size_t ra = row_of(lo), rb = row_of(hi);
        const bool cut_a = ra < rows && h0 + ra * row_bytes < lo;                              // row ra starts below lo
        const size_t in0 = cut_a ? ra + 1 : ra;
        const bool cut_b = rb < rows && rb >= in0 && h0 + rb * row_bytes < hi;                 // row rb starts inside and ends beyond hi
        hipError_t e = pageable(0, ra);
        if (e == hipSuccess && cut_a) e = one_row(ra);
        if (e == hipSuccess) e = block(in0, std::max(in0, rb));
        if (e == hipSuccess && cut_b) e = one_row(rb);
        if (e == hipSuccess) e = pageable(std::max(in0, cut_b ? rb + 1 : rb), rows);
        return e;
    }
    ~PinnedScope()
    {
        if (!pinned) return;
        // (no transfer may still be reading or writing the range when its mapping goes: the regular paths have waited already, an error
        //  return in between has not)
        if (used_stream) (void)hipStreamSynchronize(last_stream);
        const hipError_t e = hipHostUnregister(ptr);
        if (trace()) say("[vt pin] released [%p, %p): %s\n", ptr, (void*)hi, e == hipSuccess ? "ok" : hipGetErrorString(e));
        if (e != hipSuccess) (void)hipGetLastError();
    }
};

// Run the three passes X, Y, Z (reference order, transforms.py:305-307) on d_a, using d_b as the
// ping-pong partner.  Returns which buffer holds the coefficients.
int run_prefilter(float* d_a, float* d_b, int D, int H, int W, int P, bool lo_interior_axis0, hipStream_t st, float** result)
{
    float* cur = d_a;
    float* oth = d_b;
    const int order[3] = {2, 1, 0};
    int first_pass = 0;
    if (prefilter_xy_ok(D, H, W, P, cur, oth)) {
        // X and Y in one launch (vt_kernels_prefilter.hip: prefilter_xy): the plane is read once and written once for both
        VT_HIP(launch_prefilter_xy(cur, oth, D, H, W, P, st));
        float* t = cur; cur = oth; oth = t;
        first_pass = 2;
    }
    for (int i = first_pass; i < 3; ++i) {
        const int axis = order[i];
        const bool interior = (axis == 0) && lo_interior_axis0;
        if (prefilter_axis_in_place_ok(axis, D, H, W)) {
            VT_HIP(launch_prefilter_axis(axis, cur, cur, D, H, W, P, interior, st));
        } else {
            VT_HIP(launch_prefilter_axis(axis, cur, oth, D, H, W, P, interior, st));
            float* t = cur; cur = oth; oth = t;
        }
    }
    *result = cur;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// one transform call: fold offsets -> pick the resident copy and the plan -> launch -> (host output) copy back
// ---------------------------------------------------------------------------------------------------

// Which resident copy a launch samples, and what is special about its output.
struct Orientation {
    const float* src_plain = nullptr;  // plain-layout copy the launch (or its pair / quad relayout) is based on
#ifdef VT_LEGACY
    float** pair_slot = nullptr;       // where its plane-pair form lives (test build)
#endif
    float** quad_slot = nullptr;       // where its plane-quad form lives
    float** quade_slot = nullptr;      // ... and the plane-quad form of its z-convolved volume (cubic launches with an integer axis-0 offset)
    int quad_idx = 0;
    int plain_id = -1;                 // LazyCopyId of the exchanged plain-layout copy this orientation is based on (-1: the handle's own plain copy);
                                       // built by ensure_secondary_copy only when the launch -- or the relayout of a missing quad form -- reads it
    int srcD = 0, srcH = 0;            // depth / height of that copy
    int rowW = 0, rowP = 0;            // row width / pitch of that copy
    bool xswap = false;                // the kernels write an axis-0 <-> 2 exchanged result into d_tmp_x
};

bool is_marching(int kind) { return kind == 4 || kind == 5 || kind == 8; }

// src_resident = M . (d + out_plane0, h, w, 1) - (plane0, 0, 0): output-plane and resident-window offsets go into column 3
void fold_matrix(const vt_volume* v, const double m4x4[16], double m[12])
{
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 4; ++c) m[4 * r + c] = m4x4[4 * r + c];
        m[4 * r + 3] = std::fma(m4x4[4 * r], (double)v->out_plane0, m4x4[4 * r + 3]);
    }
    m[3] -= (double)v->plane0;
    for (int r = 0; r < 3; ++r) m[4 * r + 3] += (double)v->edge_pad;       // VT_EDGE_SCIPY: resident coordinate = volume coordinate + pad
}

// a handle-shaped view of the same volume with axes permuted (planning only: no buffers)
vt_volume planning_view(const vt_volume* v, int D, int H, int W, int P, int oD, int oH, int oW, bool keep_window)
{
    vt_volume sw;
    sw.dev = v->dev; sw.interp = v->interp;
    sw.D = D; sw.H = H; sw.W = W; sw.P = P;
    sw.oD = oD; sw.oH = oH; sw.oW = oW;
    sw.plane0 = keep_window ? v->plane0 : 0; sw.gD = keep_window ? v->gD : D; sw.out_plane0 = keep_window ? v->out_plane0 : 0;
    sw.lds_limit = v->lds_limit; sw.cu_count = v->cu_count; sw.tune = v->tune;
    sw.edge_pad = v->edge_pad;
    return sw;
}


// ---------------------------------------------------------------------------------------------------
// lazily built resident copies: one place that allocates, times, budgets and evicts them
// ---------------------------------------------------------------------------------------------------
// A handle that has used every orientation holds up to 13 buffers besides its plain copy.  Round 5: they are built
// through alloc_lazy(), which (1) keeps the handle inside its budget (vt_volume_set_max_resident / VT_MAX_RESIDENT_GB; the plain copy
// counts) by releasing the least recently used copies first, (2) does the same once when hipMalloc fails without any budget -- a device
// short of memory then trades an old orientation's copies for the new one instead of falling back to the slower family for good --, and
// (3) never builds a copy that cannot fit the budget at all: the caller falls back to the family that reads the plain layout.
enum LazyCopyId { kCopyT = 0, kCopyR, kCopyX, kCopyXe, kCopyQ0, kCopyQ1, kCopyQ2, kCopyQ3, kCopyQe0, kCopyQe1, kCopyQe2, kCopyQe3, kCopyTmpX, kCopyCount };

float** lazy_slot(vt_volume* v, int id)
{
    switch (id) {
        case kCopyT: return &v->d_src_t;
        case kCopyR: return &v->d_src_r;
        case kCopyX: return &v->d_src_x;
        case kCopyXe: return &v->d_src_xe;
        case kCopyQ0: return &v->d_src_q;
        case kCopyQ1: return &v->d_src_t_q;
        case kCopyQ2: return &v->d_src_r_q;
        case kCopyQ3: return &v->d_src_x_q;
        case kCopyQe0: case kCopyQe1: case kCopyQe2: case kCopyQe3: return &v->d_src_qe[id - kCopyQe0];
        default: return &v->d_tmp_x;
    }
}

uint64_t lazy_bytes(const vt_volume* v, int id)
{
    const uint64_t plain = (uint64_t)v->D * v->H * v->P * sizeof(float);
    switch (id) {
        case kCopyT: return v->d_src_t ? plain : 0;
        case kCopyR: return v->d_src_r ? (uint64_t)v->D * v->W * v->Pr * sizeof(float) : 0;
        case kCopyX: return v->d_src_x ? (uint64_t)v->W * v->H * v->Px * sizeof(float) : 0;
        case kCopyXe: return v->d_src_xe ? plain : 0;
        case kCopyQ0: case kCopyQ1: case kCopyQ2: case kCopyQ3: return v->quad_bytes[id - kCopyQ0];
        case kCopyQe0: case kCopyQe1: case kCopyQe2: case kCopyQe3: return v->quade_bytes[id - kCopyQe0];
        default: return v->d_tmp_x ? (uint64_t)v->tmp_x_elems * sizeof(float) : 0;
    }
}

uint64_t resident_now(const vt_volume* v)
{
    uint64_t tot = (uint64_t)v->D * v->H * v->P * sizeof(float) + (v->proj ? (uint64_t)3 * v->proj->H * v->proj->P * sizeof(float) : 0);
    for (int id = 0; id < kCopyCount; ++id) tot += lazy_bytes(v, id);
    return tot + v->spare_bytes;
}

void drop_spare(vt_volume* v)
{
    if (v->spare) { (void)hipFree(v->spare); (void)hipGetLastError(); }
    v->spare = nullptr; v->spare_bytes = 0;
}

void touch_lazy(vt_volume* v, int id) { if (id >= 0 && id < kCopyCount) v->copy_used[id] = v->use_clock; }

// release the least recently used lazy copy that is none of keep_a / keep_b; false when there is none
bool evict_lru(vt_volume* v, int keep_a, int keep_b)
{
    int victim = -1;
    for (int id = 0; id < kCopyCount; ++id) {
        if (id == keep_a || id == keep_b || !*lazy_slot(v, id)) continue;
        if (victim < 0 || v->copy_used[id] < v->copy_used[victim]) victim = id;
    }
    if (victim < 0) return false;
    (void)hipStreamSynchronize(v->stream);                        // no launch may still be reading what is released
    float** s = lazy_slot(v, victim);
    const size_t vbytes = (size_t)lazy_bytes(v, victim);
    drop_spare(v);
    v->spare = *s; v->spare_bytes = vbytes;                       // kept for the next build of that size (alloc_lazy); counted as resident
    *s = nullptr;
    if (victim >= kCopyQ0 && victim <= kCopyQ3) v->quad_bytes[victim - kCopyQ0] = 0;
    if (victim >= kCopyQe0 && victim <= kCopyQe3) v->quade_bytes[victim - kCopyQe0] = 0;
    if (victim == kCopyTmpX) v->tmp_x_elems = 0;
    v->copies_evicted += 1;
    if (std::getenv("VT_DEBUG_ALLOC")) std::fprintf(stderr, "[vt] evicted lazy copy %d (last used at launch %llu of %llu)\n", victim,
                                                    (unsigned long long)v->copy_used[victim], (unsigned long long)v->use_clock);
    return true;
}

// Allocate lazy copy `id` (`bytes` large) into its slot; keep_a / keep_b: copies this build reads or the launch needs.  hipSuccess, or
// hipErrorOutOfMemory when the budget or the device cannot hold it (nothing is left half-built).
// `transient`: a copy that only this build reads (the exchanged plain copy a plane-quad form is made from) and that is released right after
// the build where the budget asks for it: it does not count against the budget while the build runs -- the budget bounds what a handle
// KEEPS; for the few milliseconds of a relayout the handle may hold that source beside it (release_transient).
hipError_t alloc_lazy(vt_volume* v, int id, size_t bytes, int keep_a, int keep_b, int transient = -1)
{
    float** slot = lazy_slot(v, id);
    auto spare_fits = [&]() { return v->spare && v->spare_bytes >= bytes && v->spare_bytes - bytes <= bytes / 8; };
    if (v->max_resident) {
        const uint64_t leaving = transient >= 0 ? lazy_bytes(v, transient) : 0;
        for (;;) {
            const uint64_t held = resident_now(v) - leaving - (spare_fits() ? v->spare_bytes : 0);
            if (held + bytes <= v->max_resident) break;
            if (v->spare && !spare_fits()) { drop_spare(v); continue; }
            if (!evict_lru(v, id == keep_a ? -1 : keep_a, keep_b)) return hipErrorOutOfMemory;        // does not fit the budget at all
        }
    }
    if (spare_fits()) {
        *slot = v->spare;
        v->spare = nullptr; v->spare_bytes = 0;
        touch_lazy(v, id);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(slot), bytes);
    while (e != hipSuccess) {
        (void)hipGetLastError();
        *slot = nullptr;
        if (v->spare) drop_spare(v);
        else if (!evict_lru(v, keep_a, keep_b)) return hipErrorOutOfMemory;
        e = hipMalloc(reinterpret_cast<void**>(slot), bytes);
    }
    touch_lazy(v, id);
    return hipSuccess;
}

// after a build that read `transient`: back inside the budget, the build's source first
void release_transient(vt_volume* v, int transient, int keep)
{
    if (!v->max_resident || resident_now(v) <= v->max_resident) return;
    if (transient >= 0 && *lazy_slot(v, transient)) {
        for (int id = 0; id < kCopyCount; ++id) if (id != transient && *lazy_slot(v, id) && v->copy_used[id] == 0) v->copy_used[id] = 1;
        const uint64_t stamp = v->copy_used[transient];
        v->copy_used[transient] = 0;                                  // the least recently used one by decree
        if (!evict_lru(v, keep, -1)) v->copy_used[transient] = stamp;
    }
    while (resident_now(v) > v->max_resident) {
        if (v->spare) { drop_spare(v); continue; }
        if (!evict_lru(v, keep, -1)) break;
    }
}

// GPU time of a copy's build, for vt_volume_info.copies_ms (events of their own: the handle's timer may be running)
void lazy_build_begin(vt_volume* v)
{
    if (!v->evc0) { (void)hipEventCreate(&v->evc0); (void)hipEventCreate(&v->evc1); }
    if (v->evc0) (void)hipEventRecord(v->evc0, v->stream);
}
void lazy_build_end(vt_volume* v)
{
    v->copies_built += 1;
    if (!v->evc0 || !v->evc1) return;
    float ms = 0.f;
    if (hipEventRecord(v->evc1, v->stream) == hipSuccess && hipEventSynchronize(v->evc1) == hipSuccess &&
        hipEventElapsedTime(&ms, v->evc0, v->evc1) == hipSuccess) v->copies_ms += ms;
    (void)hipGetLastError();
}

// The exchanged plain-layout copy of an orientation (kCopyT / kCopyR / kCopyX): 0 = it exists, 1 = it cannot be built.
int ensure_lazy_plain(vt_volume* v, int id)
{
    float** slot = lazy_slot(v, id);
    if (*slot) { touch_lazy(v, id); return 0; }
    size_t bytes = 0;
    if (id == kCopyT) bytes = (size_t)v->D * v->H * v->P * sizeof(float);
    else if (id == kCopyR) { v->Pr = resident_pitch(v->H); bytes = (size_t)v->D * v->W * v->Pr * sizeof(float); }
    else if (id == kCopyX) { v->Px = resident_pitch(v->D); bytes = (size_t)v->W * v->H * v->Px * sizeof(float); }
    else return 1;
    if (alloc_lazy(v, id, bytes, id, -1) != hipSuccess) { *slot = nullptr; return 1; }
    lazy_build_begin(v);
    hipError_t e = hipSuccess;
    if (id == kCopyT) {
        e = launch_relayout_swap01(v->d_src, v->d_src_t, v->D, v->H, v->P, v->stream);
    } else {
        e = hipMemsetAsync(*slot, 0, bytes, v->stream);                                   // pad columns must be zero
        if (e == hipSuccess && id == kCopyR)        // dst[z][x][y]: element (i = y, j = z, k = x) -> (k = x, j = z, i = y)
            e = launch_transpose02(v->d_src, v->d_src_r, v->H, v->D, v->W, v->P, (int64_t)v->H * v->P, v->Pr, (int64_t)v->W * v->Pr, v->stream);
        else if (e == hipSuccess)
            e = launch_transpose02(v->d_src, v->d_src_x, v->D, v->H, v->W, (int64_t)v->H * v->P, v->P, (int64_t)v->H * v->Px, v->Px, v->stream);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(v->stream);
        (void)hipFree(*slot);
        (void)hipGetLastError();
        *slot = nullptr;
        return 1;
    }
    lazy_build_end(v);
    return 0;
}

// Rotations about axis 1 ([a 0 b; 0 1 0; c 0 d]): the same problem with axes 0 and 1 exchanged is axis-0-separable.
// A second resident copy with those axes exchanged (built once, lazily) lets the marching kernels serve it; only
// the output addressing changes (plane stride oW, row stride oH*oW).  Whole-volume handles only (no slab offsets).
int try_axis1_exchange(vt_volume* v, const double m[12], int flags, size_t n_out, AffineParams* p, TilePlan* plan, Orientation* ori)
{
    const bool ysep = !(flags & (VT_NO_ZSEP | VT_NO_MARCH | VT_FORCE_DIRECT)) && m[5] == 1.0 && m[4] == 0.0 && m[6] == 0.0 &&
                      m[1] == 0.0 && m[9] == 0.0 && std::fabs(m[7]) < 1.0e9 &&
                      !(m[0] == 1.0 && m[2] == 0.0 && m[8] == 0.0) &&
                      v->plane0 == 0 && v->out_plane0 == 0 && v->gD == v->D && v->D <= 65535 && v->H <= 65535 &&
                      (n_out >= (size_t)64 * 64 * 64 || (flags & VT_FORCE_TILED));
    if (!ysep) return 0;
    vt_volume sw = planning_view(v, v->H, v->D, v->W, v->P, v->oH, v->oD, v->oW, false);
    const int pi[3] = {1, 0, 2};
    double ms[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) ms[4 * r + c] = m[4 * pi[r] + pi[c]];
        ms[4 * r + 3] = m[4 * pi[r] + 3];
    }
    AffineParams ps;
    std::memset(&ps, 0, sizeof(ps));
    TilePlan plans;
    plan_launch(&sw, ms, flags, &ps, &plans);
    if (!is_marching(plans.kind)) return 0;
    *p = ps; *plan = plans;
    p->ostride = v->oW; p->orow = (int64_t)v->oH * v->oW;
    p->ord[0] = 1; p->ord[1] = 0; p->ord[2] = 2;          // original (d, h, w) = this launch's columns (1, 0, 2)
    ori->src_plain = v->d_src_t; ori->plain_id = kCopyT; ori->quad_slot = &v->d_src_t_q; ori->quade_slot = &v->d_src_qe[1]; ori->quad_idx = 1;
#ifdef VT_LEGACY
    ori->pair_slot = &v->d_src_t_zp;
#endif
    ori->srcD = v->H; ori->srcH = v->D;
    return 0;
}

// Rotations about axis 2 ([a b 0; c d 0; 0 0 1]): axis-0-separable after exchanging axes 0 and 2.  The marching kernels
// run on the exchanged copy and produce an exchanged result, which one transpose pass (8 B/voxel) turns back.
// Cubic only: trilinear rotations about axis 2 are the bounding-box kernel's best case (x stays contiguous: 0.34 ms at
// 512^3, faster than marching 0.25 + transposing 0.2); VT_FORCE_XSWAP (diagnostic) takes the exchange path regardless.
int try_axis2_exchange(vt_volume* v, const double m[12], int flags, size_t n_out, AffineParams* p, TilePlan* plan, Orientation* ori)
{
    const bool xsep = !(flags & (VT_NO_ZSEP | VT_NO_MARCH | VT_FORCE_DIRECT | VT_KEEP_OUTSIDE)) &&
                      (is_cubic(v->interp) || (flags & VT_FORCE_XSWAP)) &&
                      m[10] == 1.0 && m[8] == 0.0 && m[9] == 0.0 && m[2] == 0.0 && m[6] == 0.0 && std::fabs(m[11]) < 1.0e9 &&
                      !(m[0] == 1.0 && m[1] == 0.0 && m[4] == 0.0) && !(m[5] == 1.0 && m[1] == 0.0 && m[4] == 0.0) &&
                      v->plane0 == 0 && v->out_plane0 == 0 && v->gD == v->D && v->H <= 65535 && v->oH <= 65535 &&
                      (n_out >= (size_t)64 * 64 * 64 || (flags & VT_FORCE_TILED));
    if (!xsep) return 0;
    vt_volume sw = planning_view(v, v->W, v->H, v->D, resident_pitch(v->D), v->oW, v->oH, v->oD, false);
    const int pi[3] = {2, 1, 0};
    double ms[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) ms[4 * r + c] = m[4 * pi[r] + pi[c]];
        ms[4 * r + 3] = m[4 * pi[r] + 3];
    }
    AffineParams ps;
    std::memset(&ps, 0, sizeof(ps));
    TilePlan plans;
    plan_launch(&sw, ms, flags, &ps, &plans);
    if (!is_marching(plans.kind)) return 0;
    v->Px = sw.P;
    *p = ps; *plan = plans;
    p->ord[0] = 2; p->ord[1] = 1; p->ord[2] = 0;          // original (d, h, w) = this launch's columns (2, 1, 0)
    ori->src_plain = v->d_src_x; ori->plain_id = kCopyX; ori->quad_slot = &v->d_src_x_q; ori->quade_slot = &v->d_src_qe[3]; ori->quad_idx = 3;
#ifdef VT_LEGACY
    ori->pair_slot = &v->d_src_x_zp;
#endif
    ori->srcD = v->W; ori->srcH = v->H; ori->rowW = v->D; ori->rowP = v->Px;
    ori->xswap = true;
    return 0;
}

// In-plane maps closer to a quarter turn than to the identity (|m12| > |m11|: rotations about axis 0 by 45..135 and
// 225..315 degrees): sampled from an in-plane TRANSPOSED resident copy the same map has its rows 1 and 2 exchanged and
// falls into the 0..45 degree class, whose footprints are wide in x (long staged rows, lanes walk along LDS rows
// instead of down a column).  Only the source side changes; the output is written as usual.
int try_inplane_transposed(vt_volume* v, const double m[12], int flags, size_t n_out, AffineParams* p, TilePlan* plan, Orientation* ori)
{
    const bool zsep_m = m[0] == 1.0 && m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[8] == 0.0 && std::fabs(m[3]) < 1.0e9;
    const bool rsep = zsep_m && !(flags & (VT_NO_ZSEP | VT_NO_MARCH | VT_FORCE_DIRECT | VT_NO_RSWAP)) &&
                      std::fabs(m[6]) > std::fabs(m[5]) && std::fabs(m[9]) > std::fabs(m[10]) && v->D <= 65535 &&
                      (n_out >= (size_t)64 * 64 * 64 || (flags & VT_FORCE_TILED));
    if (!rsep) return 0;
    vt_volume sw = planning_view(v, v->D, v->W, v->H, resident_pitch(v->H), v->oD, v->oH, v->oW, true);
    double ms[12];
    for (int c = 0; c < 4; ++c) { ms[c] = m[c]; ms[4 + c] = m[8 + c]; ms[8 + c] = m[4 + c]; }
    AffineParams ps;
    std::memset(&ps, 0, sizeof(ps));
    TilePlan plans;
    plan_launch(&sw, ms, flags, &ps, &plans);
    if (!is_marching(plans.kind)) return 0;
    v->Pr = sw.P;
    *p = ps; *plan = plans;
    ori->src_plain = v->d_src_r; ori->plain_id = kCopyR; ori->quad_slot = &v->d_src_r_q; ori->quade_slot = &v->d_src_qe[2]; ori->quad_idx = 2;
#ifdef VT_LEGACY
    ori->pair_slot = &v->d_src_r_zp;
#endif
    ori->srcD = v->D; ori->srcH = v->W; ori->rowW = v->H; ori->rowP = v->Pr;
    // Tile order on the transposed copy: h fastest for the round-1 marching kernels (consecutive tiles read neighbouring source
    // rows); the plane-quad kernel with its 2-D grid keeps w fastest -- [measured, angles 50..130] 1024^3 trilinear 1.58-1.70 ->
    // 1.51-1.63 ms, 1024^3 cubic at 80 / 90 degrees 1.76 / 1.79 -> 1.65 / 1.61, 512^3 equal or better (consecutive tiles complete whole
    // output rows instead of writing down a column of tiles).  VT_RSWAP_WFAST=1 / 0 forces either.
    const bool wfast = v->tune.rswap_wfast >= 0 ? v->tune.rswap_wfast != 0 : plans.kind == 8;
    if (!wfast) p->flags |= (1 << 24);
    return 0;
}

// General matrices (none of the forms above): the general-matrix kernels stage source ROWS -- runs along the resident copy's fastest axis --
// and their lanes walk along the output's w axis.  Where w follows source axis 0 or 1 more closely than axis 2, the rows a tile touches
// are many and short on the plain copy (partly used cache lines, short staged vectors, lanes of a wave scattered over LDS rows); sampled
// from the copy whose FASTEST axis is the one w follows ([x][y][z] of the axis-2 exchange, [z][x][y] of the in-plane transposition: the
// copies the marching kernels use) the same launch has long rows.  Only the source side changes -- the matrix rows are permuted with the
// copy's axes, the output is written as usual, the skirt test's chains are per row and unchanged.  [measured, 512^3, the reference's 100
// random rotations, tools/general_reorient_probe.py] trilinear 0.449 -> 0.426 ms, cubic 1.004 -> 0.956 (best of the three copies per
// matrix: 0.422 / 0.949).  A copy is built at the handle's FOURTH call that asks for it: a handle used once or twice (vt_affine_oneshot)
// never pays a transpose pass for a few per cent of one launch.
int try_general_reorient(vt_volume* v, const double m[12], int flags, size_t n_out, AffineParams* p, TilePlan* plan, Orientation* ori)
{
    if ((flags & (VT_FORCE_DIRECT | VT_NO_REORIENT)) || v->tune.reorient == 0) return 0;
    if (n_out < (size_t)192 * 192 * 192 && !(flags & VT_FORCE_TILED)) return 0;      // launch-bound sizes: nothing to win
    // the source axis the output's w direction follows (the plain copy keeps ties and near-ties: no copy, no churn)
    const double c[3] = {std::fabs(m[2]), std::fabs(m[6]), std::fabs(m[10])};
    int a = 2;
    if (c[1] > 1.15 * c[2] && c[1] >= c[0]) a = 1;
    if (c[0] > 1.15 * c[2] && c[0] > c[1]) a = 0;
    if (a == 2) return 0;
    if (a == 0 && !(v->plane0 == 0 && v->out_plane0 == 0 && v->gD == v->D)) return 0;  // slab windows live on axis 0: it stays the slowest
    float** const slot = (a == 1) ? &v->d_src_r : &v->d_src_x;
    int* const asked = &v->reorient_asked[a];
    if (!*slot && ++*asked < ((flags & VT_FORCE_TILED) ? 1 : v->tune.reorient)) return 0;
    const vt_volume sw = (a == 1) ? planning_view(v, v->D, v->W, v->H, resident_pitch(v->H), v->oD, v->oH, v->oW, true)
                                  : planning_view(v, v->W, v->H, v->D, resident_pitch(v->D), v->oD, v->oH, v->oW, false);
    const int pi[3] = {a == 1 ? 0 : 2, a == 1 ? 2 : 1, a == 1 ? 1 : 0};                // source axis of the copy's axis r
    double ms[12];
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 4; ++k) ms[4 * r + k] = m[4 * pi[r] + k];
    AffineParams ps;
    std::memset(&ps, 0, sizeof(ps));
    TilePlan plans;
    plan_launch(&sw, ms, flags, &ps, &plans);
    if (!(plans.kind == 2 || plans.kind == 6 || plans.kind == 9)) return 0;           // the general-matrix kernels only
    if (a == 1) v->Pr = sw.P; else v->Px = sw.P;
    if (ensure_lazy_plain(v, a == 1 ? kCopyR : kCopyX)) {
        *asked = -64;                         // no room for another copy: the plain layout serves this matrix and the next 64 requests
        return 0;
    }
    *p = ps; *plan = plans;
    ori->src_plain = *slot; ori->plain_id = a == 1 ? kCopyR : kCopyX;
    ori->srcD = sw.D; ori->srcH = sw.H; ori->rowW = sw.W; ori->rowP = sw.P;
    return 0;
}

// Maps that leave axis 2 alone with an integer offset (rotations about axis 2 through the default centre): the row kernel on the plain
// copy (trilinear) or on its x-convolved form (cubic; built lazily, one relayout pass).  vt_kernels_rows.hip.
int try_rows(vt_volume* v, const double m[12], int flags, AffineParams* p, TilePlan* plan, Orientation* ori)
{
    AffineParams ps;
    std::memset(&ps, 0, sizeof(ps));
    TilePlan plans = TilePlan();
    plans.kind = 0;
    if (!plan_rows(v, m, flags, &ps, &plans) || plans.kind != 10) return 0;
    const bool use_xe = is_cubic(v->interp) && !(ps.flags & (1 << 16));      // (a fractional axis-2 offset forms its own x-sums on the plain copy)
    if (use_xe && !v->d_src_xe) {
        if (v->xe_retry_in > 0) { --v->xe_retry_in; return 0; }
        const size_t bytes = (size_t)v->D * v->H * v->P * sizeof(float);
        if (alloc_lazy(v, kCopyXe, bytes, kCopyXe, -1) != hipSuccess) {
            v->d_src_xe = nullptr;            // no room for the copy: the exchange path or the general kernels serve this matrix
            v->xe_retry_in = 64;
            return 0;
        }
        lazy_build_begin(v);
        const bool simple = v->interp == VT_BSPLINE_SIMPLE || v->interp == VT_FILT_BSPLINE_SIMPLE;
        if (launch_relayout_xfir(v->d_src, v->d_src_xe, v->D, v->H, v->W, v->P, simple, v->stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(v->stream);
            (void)hipFree(v->d_src_xe);
            v->d_src_xe = nullptr;
            v->xe_retry_in = 64;
            return 0;
        }
        lazy_build_end(v);
    }
    *p = ps; *plan = plans;
    if (use_xe) touch_lazy(v, kCopyXe);
    ori->src_plain = use_xe ? v->d_src_xe : v->d_src;
    return 0;
}

void note_launch(vt_volume* v, int kind, const TilePlan& plan, const AffineParams& p, size_t n_out)
{
    v->last_kernel = kind;
    const bool tiled = kind >= 2;
    v->last_tile[0] = tiled ? plan.td : 0; v->last_tile[1] = tiled ? plan.th : 0; v->last_tile[2] = tiled ? plan.tw : 0;
    v->last_lds[0] = tiled ? p.Lz : 0; v->last_lds[1] = tiled ? p.Ly : 0; v->last_lds[2] = tiled ? p.Lx : 0;
    v->last_lds_bytes = tiled ? plan.lds_bytes : 0;
    v->last_grid = tiled ? plan.grid : (int)((n_out + 255) / 256);
}

// Build the secondary resident copy a plan needs (once per handle and orientation).  Returns 0 when the copy exists, 1 when it
// could not be built -- no device memory for it, or a relayout launch that was refused: the slot is freed and cleared (a zero-filled
// copy must never survive: later calls would sample it silently) and the caller re-plans without this kernel family --, < 0 never.
int ensure_secondary_copy(vt_volume* v, const TilePlan& plan, const AffineParams& p, Orientation& ori)
{
    float** slot = nullptr;
    size_t bytes = 0;
    int id = -1;
    const bool zfir = plan.kind == 8 && (p.flags & (1 << 19)) != 0;       // the z-convolved copy (vt_plan.hip: plan_quad)
    if (plan.kind == 8) {
        slot = zfir ? ori.quade_slot : ori.quad_slot;
        id = (zfir ? kCopyQe0 : kCopyQ0) + ori.quad_idx;
        bytes = (size_t)((ori.srcD + 3) / 4) * ori.srcH * p.sPq * sizeof(float);
#ifdef VT_LEGACY
    } else if (plan.kind == 5) {
        slot = ori.pair_slot;
        bytes = (size_t)((ori.srcD + 1) / 2) * ori.srcH * p.sP2 * sizeof(float);
#endif
    }
    // the exchanged result buffer of the axis-0 <-> 2 orientation
    if (ori.xswap) {
        const size_t n_out = (size_t)v->oD * v->oH * v->oW;
        if (v->tmp_x_elems < n_out) {
            if (v->d_tmp_x) { (void)hipStreamSynchronize(v->stream); (void)hipFree(v->d_tmp_x); v->d_tmp_x = nullptr; v->tmp_x_elems = 0; }
            if (alloc_lazy(v, kCopyTmpX, n_out * sizeof(float), kCopyTmpX, id) != hipSuccess) { v->d_tmp_x = nullptr; return 1; }
            v->tmp_x_elems = n_out;
        }
        touch_lazy(v, kCopyTmpX);
    }
    if (slot && *slot) { touch_lazy(v, id); return 0; }                    // the launch reads this form only: its plain-layout source may be gone
    // The exchanged plain-layout copy of the orientation: read by the launch itself (kinds that sample the plain layout) or by the relayout
    // that builds the missing plane-quad form.  Built here and not when the orientation is chosen: under a budget it is the first copy
    // evicted once its quad form exists, and a sweep must not rebuild it on every call.
    // (Round 5: the plane-quad forms of the in-plane transposed orientation come straight from the handle's own plain copy --
    // relayout_zquad_swap12 -- when that exchanged copy does not exist yet: nothing else reads it on this path.)
    const bool fused_swap12 = plan.kind == 8 && ori.plain_id == kCopyR && !v->d_src_r && !v->tune.no_fused_relayout;
    if (ori.plain_id >= 0 && !fused_swap12) {
        if (ensure_lazy_plain(v, ori.plain_id)) return 1;
        ori.src_plain = *lazy_slot(v, ori.plain_id);
    }
    if (!slot) return 0;
#ifdef VT_LEGACY
    if (v->tune.test_fail_copy) return 1;      // VT_TEST_FAIL_COPY (test build): the allocation-failure path, for tests/test_gpu_parity.py
#endif
    // A copy that did not fit is not attempted again at once: a device that is short of memory would pay a volume-sized hipMalloc
    // (and its failure) on every call.  The next attempt comes kCopyRetryCalls calls later.
    constexpr int kCopyRetryCalls = 64;
    int* const retry = (plan.kind == 8) ? &v->copy_retry_in[ori.quad_idx + (zfir ? 4 : 0)] : nullptr;
    if (retry && *retry > 0) { --*retry; return 1; }
    hipError_t e = hipSuccess;
    if (id >= 0) e = alloc_lazy(v, id, bytes, id, fused_swap12 ? -1 : ori.plain_id, fused_swap12 ? -1 : ori.plain_id);
    else e = hipMalloc(reinterpret_cast<void**>(slot), bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();              // no room for another copy of the volume: a family that reads the plain layout serves the call
        *slot = nullptr;
        if (retry) *retry = kCopyRetryCalls;
        return 1;
    }
    lazy_build_begin(v);
    if (plan.kind != 8) e = hipMemsetAsync(*slot, 0, bytes, v->stream);       // positions beyond the row's width stay zero (the plane-quad relayouts write them themselves)
    if (e == hipSuccess) {
        if (fused_swap12)
            e = launch_relayout_zquad_swap12(v->d_src, *slot, v->D, v->H, v->W, v->P, p.sPq, zfir, (p.flags & (1 << 18)) != 0, v->stream);
        else if (zfir)
            e = launch_relayout_zquad_fir(ori.src_plain, *slot, ori.srcD, ori.srcH, ori.rowW, ori.rowP, p.sPq, (p.flags & (1 << 18)) != 0, v->stream);
        else if (plan.kind == 8)
            e = launch_relayout_zquad(ori.src_plain, *slot, ori.srcD, ori.srcH, ori.rowW, ori.rowP, p.sPq, v->stream);
#ifdef VT_LEGACY
        else
            e = launch_relayout_zpair(ori.src_plain, *slot, ori.srcD, ori.srcH, ori.rowW, ori.rowP, p.sP2, v->stream);
#endif
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(v->stream);                       // nothing may still be writing the buffer that is freed
        (void)hipFree(*slot);
        (void)hipGetLastError();
        *slot = nullptr;
        if (retry) *retry = kCopyRetryCalls;
        return 1;
    }
    lazy_build_end(v);
    if (zfir) v->quade_bytes[ori.quad_idx] = bytes;
    else if (plan.kind == 8) v->quad_bytes[ori.quad_idx] = bytes;
    if (id >= 0) release_transient(v, fused_swap12 ? -1 : ori.plain_id, id);             // (lazy_build_end has waited for the relayout: its source may go)
    if (std::getenv("VT_DEBUG_ALLOC")) std::fprintf(stderr, "[vt] secondary copy kind %d orientation %d at %p, %zu bytes (plain source %p)\n", plan.kind, ori.quad_idx, (void*)*slot, bytes, (const void*)ori.src_plain);
    if (zfir) v->quade_bytes[ori.quad_idx] = bytes;
    else if (plan.kind == 8) v->quad_bytes[ori.quad_idx] = bytes;
#ifdef VT_LEGACY
    else v->P2 = p.sP2;
#endif
    return 0;
}

// launch the plan's kernel into d_out (the secondary copy it reads exists: ensure_secondary_copy)
int launch_planned(vt_volume* v, const TilePlan& plan, const AffineParams& p, const Orientation& ori, float* d_out, size_t n_out)
{
    if (plan.kind == 8) {
        float* const srcq = (p.flags & (1 << 19)) ? *ori.quade_slot : *ori.quad_slot;
        if (!srcq) return fail(VT_EINVAL, "internal: plane-quad copy missing");
        // Every other launch of a handle walks the chunk layers from the last to the first: the source planes the previous launch
        // read last are still in the memory-side cache (256 MB, it sees reads and writes alike) when this one starts with them.
        // A schedule only: each workgroup computes what it computed before.  (VT_QUAD_PINGPONG=0: always first to last; 2: always
        // last to first -- the control, which measures like 0.)
        AffineParams q = p;
        if (v->tune.quad_pingpong == 2 || (v->tune.quad_pingpong == 1 && ((v->launch_no++) & 1))) q.flags |= (1 << 20);
        // (Round 5 measured per-launch TILE TABLES here -- the in-plane set-up of every tile worked out once, by a one-layer pass of the same
        //  kernel in front of the real launch, and read back by every chunk layer: bit-identical, the real launch faster and flatter over the
        //  angles, but the pass itself sits in front of every launch with 12-14 us, twice what it saves: 512^3 filt_bspline sweep 0.1983 ms
        //  against 0.1915, 1024^3 1.586 against 1.565, and the second code path cost the default one 1 %.  profiles/r05_tile_tables.txt;
        //  not in the tree.)
        VT_HIP(launch_affine_quad(plan.cfg, v->interp, srcq, d_out, q, plan.grid, plan.lds_bytes, v->stream));
#ifdef VT_LEGACY
    } else if (plan.kind == 5) {
        if (!*ori.pair_slot) return fail(VT_EINVAL, "internal: plane-pair copy missing");
        VT_HIP(launch_affine_zpair(plan.cfg, v->interp, *ori.pair_slot, d_out, p, plan.grid, plan.lds_bytes, v->stream));
#endif
    } else if (plan.kind == 9 || plan.kind == 6) {
        if (!v->d_queue) {                                           // tile counters of the persistent kernels: zero between launches
            VT_HIP(hipMalloc(reinterpret_cast<void**>(&v->d_queue), 9 * 128));
            VT_HIP(hipMemsetAsync(v->d_queue, 0, 9 * 128, v->stream));
        }
        if (plan.kind == 9) {
#ifdef VT_EXPERIMENTS      // occupancy experiment: fewer resident workgroups per CU through a larger LDS request / a smaller persistent grid
            static const int exp_lds = std::getenv("VT_EXP_BLOCK_LDS") ? std::atoi(std::getenv("VT_EXP_BLOCK_LDS")) : 0;
            static const int exp_grid = std::getenv("VT_EXP_BLOCK_GRID") ? std::atoi(std::getenv("VT_EXP_BLOCK_GRID")) : 0;
            VT_HIP(launch_affine_block(plan.cfg, plan.th, v->interp, ori.src_plain, d_out, v->d_zeros, v->d_queue, p, plan.geo, exp_grid > 0 ? exp_grid : plan.grid,
                                       std::max(plan.lds_bytes, exp_lds), v->stream));
#else
            VT_HIP(launch_affine_block(plan.cfg, plan.th, v->interp, ori.src_plain, d_out, v->d_zeros, v->d_queue, p, plan.geo, plan.grid, plan.lds_bytes, v->stream));
#endif
        }
        else if (!is_cubic(v->interp) && v->tune.span != 0)
            VT_HIP(launch_affine_span(plan.cfg, ori.src_plain, d_out, v->d_zeros, v->d_queue, p, plan.geo, plan.grid, plan.lds_bytes, v->stream));
        else
            VT_HIP(launch_affine_packed(plan.cfg, v->interp, ori.src_plain, d_out, v->d_zeros, v->d_queue, p, plan.geo, plan.grid, plan.lds_bytes, v->stream));
#ifdef VT_LEGACY
    } else if (plan.kind == 4) {
        VT_HIP(launch_affine_march(plan.cfg, v->interp, ori.src_plain, d_out, p, plan.grid, plan.lds_bytes, v->stream));
#endif
    } else if (plan.kind == 10) {
        VT_HIP(launch_affine_rows(v->interp, plan.td, ori.src_plain, d_out, v->d_zeros, p, plan.lds_bytes, v->stream));
    } else if (plan.kind >= 2) {
        VT_HIP(launch_affine_tiled(plan.cfg, v->interp, plan.kind == 3, ori.src_plain, d_out, v->d_zeros, p, plan.grid, plan.lds_bytes, v->stream));
    } else {
        VT_HIP(launch_affine_direct(v->interp, v->d_src, d_out, p, v->stream));
    }
    note_launch(v, plan.kind >= 2 ? plan.kind : 1, plan, p, n_out);
    return 0;
}

// device staging buffer for a host `out` (recycled through the per-device cache)
int host_output_buffer(vt_volume* v, size_t n_elems, float** d_out)
{
    if (v->scratch_elems < n_elems) {
        if (v->d_scratch_out) { cached_free(v->dev, v->d_scratch_out, v->scratch_elems * sizeof(float)); v->d_scratch_out = nullptr; v->scratch_elems = 0; }
        VT_HIP(cached_malloc(v->dev, reinterpret_cast<void**>(&v->d_scratch_out), n_elems * sizeof(float)));
        v->scratch_elems = n_elems;
    }
    *d_out = v->d_scratch_out;
    return 0;
}

int do_affine(vt_volume* v, const double m4x4[16], float* out, int flags)
{
    if (!v || !m4x4 || !out) return fail(VT_EINVAL, "NULL argument");
    if (v->deferred) return fail(VT_EINVAL, "handle was created with VT_SRC_DEFERRED and has not been finalized (vt_volume_finalize)");
    int rc = use_device(v->dev);
    if (rc) return rc;
    (void)hipGetLastError();                      // a stale sticky error of an unrelated call must not fail this launch's check
    double m[12];
    fold_matrix(v, m4x4, m);
    for (int i = 0; i < 12; ++i)
        if (!std::isfinite(m[i])) return fail(VT_EINVAL, "matrix entry %d is not finite", i);

    AffineParams p;
    TilePlan plan;
    Orientation ori;
    const size_t n_out = (size_t)v->oD * v->oH * v->oW;
    v->use_clock += 1;
    // Plan, then make sure the resident copy the plan reads exists.  A copy that cannot be built (device memory: a handle that has
    // used every orientation holds up to 8 copies of its volume) takes its kernel family out of the running and the call is
    // planned again: quad -> pair (cubic) / plain marching -> ... every family from kind 4 down reads the plain layout.
    int deny = 0;
    for (int attempt = 0;; ++attempt) {
        std::memset(&p, 0, sizeof(p));
        plan = TilePlan();
        plan.kind = 0;
        ori = Orientation();
        ori.src_plain = v->d_src; ori.quad_slot = &v->d_src_q; ori.quade_slot = &v->d_src_qe[0]; ori.quad_idx = 0;
#ifdef VT_LEGACY
        ori.pair_slot = &v->d_src_zp;
#endif
        ori.srcD = v->D; ori.srcH = v->H; ori.rowW = v->W; ori.rowP = v->P;
        const int pf = flags | deny;
        // single-axis rotations about axes 1 / 2 and in-plane maps near a quarter turn march on an exchanged resident copy
        if ((rc = try_rows(v, m, pf, &p, &plan, &ori))) return rc;
        if (plan.kind == 0 && (rc = try_axis1_exchange(v, m, pf, n_out, &p, &plan, &ori))) return rc;
        if (plan.kind == 0 && (rc = try_axis2_exchange(v, m, pf, n_out, &p, &plan, &ori))) return rc;
        if (plan.kind == 0 && (rc = try_inplane_transposed(v, m, pf, n_out, &p, &plan, &ori))) return rc;
        if (plan.kind == 0 && (rc = try_general_reorient(v, m, pf, n_out, &p, &plan, &ori))) return rc;
        if (plan.kind == 0) plan_launch(v, m, pf, &p, &plan);
        const int miss = ensure_secondary_copy(v, plan, p, ori);
        if (miss == 0) break;
        if (attempt >= 3) return fail(VT_EINVAL, "no kernel family can serve this call (secondary resident copies cannot be built)");
        // (the z-convolved copy of KIND 4 missing: the four-plane kernel on the plain plane-quad copy is next in line; a general-matrix
        //  launch whose axis-permuted plain copy could not be built samples the plain copy)
        if (plan.kind == 8) deny |= ((p.flags & (1 << 19)) && !(deny & VT_NO_ZFIR)) ? VT_NO_ZFIR : VT_NO_QUAD;
        else if (plan.kind == 2 || plan.kind == 6 || plan.kind == 9) deny |= VT_NO_REORIENT;
        else deny |= VT_NO_ZPAIR;
    }

    float* d_out = out;
    const bool host_out = !(flags & VT_OUT_DEVICE);
    PinnedScope pin(host_out ? out : nullptr, host_out ? n_out * sizeof(float) : 0);       // the caller's array: in (keep_outside) and out
    if (host_out) {
        if ((rc = host_output_buffer(v, n_out, &d_out))) return rc;
        if (flags & VT_KEEP_OUTSIDE)   // caller's stale values must survive: bring them in first
            VT_HIP(pin.copy(d_out, out, n_out * sizeof(float), hipMemcpyHostToDevice, v->stream));
    }
    float* const d_final = d_out;
    if (ori.xswap) d_out = v->d_tmp_x;            // the kernels write the exchanged result [w][h][d]
    if ((rc = launch_planned(v, plan, p, ori, d_out, n_out))) return rc;
    if (ori.xswap)                                // [w][h][d] -> [d][h][w]
        VT_HIP(launch_transpose02(v->d_tmp_x, d_final, v->oW, v->oH, v->oD, (int64_t)v->oH * v->oD, v->oD,
                                  (int64_t)v->oH * v->oW, v->oW, v->stream));
    if (host_out) {
        VT_HIP(pin.copy(out, d_final, n_out * sizeof(float), hipMemcpyDeviceToHost, v->stream));
        VT_HIP(hipStreamSynchronize(v->stream));
    }
    return 0;
}

constexpr int kSrcNone = 1 << 30;   // internal create flag: no source data (zero-filled resident buffer)

// Last step of building a handle: the one-time prefilter of filt_* interpolations over the resident samples
// (transforms.py:195-197, volume.py:48-50), then the handle is usable.
int finalize_resident(vt_volume* v, bool lo_interior)
{
    const int dev = v->dev;
    if (is_filtered(v->interp)) {
        const size_t bytes = v->src_bytes;
        float* d_tmp = nullptr;
        VT_HIP(cached_malloc(dev, reinterpret_cast<void**>(&d_tmp), bytes));
        {
            hipError_t em = hipMemset2DAsync(d_tmp + v->W, (size_t)v->P * sizeof(float), 0, (size_t)(v->P - v->W) * sizeof(float),
                                             (size_t)v->D * v->H, v->stream);
            if (em != hipSuccess) { cached_free(dev, d_tmp, bytes); return fail((int)em, "memset: %s", hipGetErrorString(em)); }
        }
        float* res = nullptr;
        hipEventRecord(v->ev0, v->stream);
        int rc = run_prefilter(v->d_src, d_tmp, v->D, v->H, v->W, v->P, lo_interior, v->stream, &res);
        hipEventRecord(v->ev1, v->stream);
        hipError_t es = hipStreamSynchronize(v->stream);
        if (rc || es != hipSuccess) {
            cached_free(dev, d_tmp, bytes);
            if (!rc) rc = fail((int)es, "prefilter: %s", hipGetErrorString(es));
            return rc;
        }
        hipEventElapsedTime(&v->prefilter_ms, v->ev0, v->ev1);
        if (res == d_tmp) { cached_free(dev, v->d_src, bytes); v->d_src = d_tmp; }
        else cached_free(dev, d_tmp, bytes);
    } else {
        VT_HIP(hipStreamSynchronize(v->stream));
    }
    v->deferred = false;
    v->proj_sum_valid = false;
    return 0;
}

int create_common(int dev, int D, int H, int W, int interp, const float* data, int cflags,
                  int64_t plane0, int64_t gD, int64_t out_plane0, int oD, vt_volume_t** out)
{
    if (!out) return fail(VT_EINVAL, "NULL handle pointer");
    *out = nullptr;
    if (!data && !(cflags & (kSrcNone | VT_SRC_DEFERRED))) return fail(VT_EINVAL, "NULL data pointer");
    if (D <= 0 || H <= 0 || W <= 0 || oD <= 0) return fail(VT_EINVAL, "non-positive dims (%d,%d,%d) out depth %d", D, H, W, oD);
    if (interp < VT_LINEAR || interp > VT_FILT_BSPLINE_SIMPLE) return fail(VT_EINVAL, "unknown interpolation code %d", interp);
    if ((int64_t)D * H > 0x7fffffffLL || (int64_t)H * W > 0x7fffffffLL) return fail(VT_EUNSUPPORTED, "plane count/size exceeds 2^31");
    if (gD <= 0 || plane0 + D < 0 || plane0 > gD) return fail(VT_EINVAL, "slab window [%lld,%lld) outside global depth %lld",
                                                               (long long)plane0, (long long)(plane0 + D), (long long)gD);
    int rc = init_device(dev);
    if (rc) return rc;

    const int uD = D, uH = H, uW = W;              // the caller's dims; D, H, W below are the resident copy's
    int pad = 0;
    if (cflags & VT_EDGE_SCIPY) {
        if (plane0 != 0 || gD != D || out_plane0 != 0 || oD != D || (cflags & (kSrcNone | VT_SRC_DEFERRED)))
            return fail(VT_EUNSUPPORTED, "VT_EDGE_SCIPY is available for whole-volume handles only");
        // one mirrored voxel serves every in-range tap; the prefiltered interpolations carry 16, so that the ordinary prefilter's
        // own boundary treatment sits 16 samples away from the data (|z|^16 = 7e-10) and what reaches the data is the
        // mirror-boundary solution scipy computes
        pad = is_filtered(interp) ? 16 : 1;
        D += 2 * pad; H += 2 * pad; W += 2 * pad;
        gD = D;
        if ((int64_t)D * H > 0x7fffffffLL || (int64_t)H * W > 0x7fffffffLL) return fail(VT_EUNSUPPORTED, "plane count/size exceeds 2^31");
    }
    vt_volume* v = new (std::nothrow) vt_volume();
    if (!v) return fail(VT_ENOMEM, "out of host memory");
    v->dev = dev; v->interp = interp; v->D = D; v->H = H; v->W = W;
    v->oD = oD; v->oH = uH; v->oW = uW;
    v->plane0 = plane0; v->gD = gD; v->out_plane0 = out_plane0;
    v->edge_pad = pad;
    v->tune.read();
    if (v->tune.max_resident_gb > 0.0) v->max_resident = (uint64_t)(v->tune.max_resident_gb * 1073741824.0);

    auto cleanup = [&](int code) {
        vt_volume_destroy(v);
        return code;
    };
    if (dev < 64) {
        v->cu_count = g_cu_count[dev];
        v->lds_limit = g_lds_limit[dev];
        v->d_zeros = g_zeros[dev];
    } else {
        return cleanup(fail(VT_ENODEV, "device index %d beyond the supported 64", dev));
    }

#define VT_HIPC(call)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return cleanup(fail((int)e_, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__)); \
    } while (0)

    VT_HIPC(cached_stream(dev, &v->stream));      // blocking stream (see cached_stream)
    VT_HIPC(cached_event(dev, &v->ev0));
    VT_HIPC(cached_event(dev, &v->ev1));
    // resident layout: rows padded to a multiple of 4 floats so every row starts 16-byte aligned (the tiled
    // kernel stages with 16-byte direct-to-LDS loads); pad columns are zero = the border value
    v->P = resident_pitch(W);
    const size_t bytes = (size_t)D * H * v->P * sizeof(float);
    VT_HIPC(cached_malloc(dev, reinterpret_cast<void**>(&v->d_src), bytes));
    v->src_bytes = bytes;
    const hipMemcpyKind kind = (cflags & VT_SRC_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (cflags & (kSrcNone | VT_SRC_DEFERRED)) {
        // internal helper volumes (written by a kernel later) and deferred handles (filled by vt_volume_upload_planes, made
        // usable by vt_volume_finalize): zero-filled -- planes that are never uploaded read as the border colour
        VT_HIPC(hipMemsetAsync(v->d_src, 0, bytes, v->stream));
        VT_HIPC(hipStreamSynchronize(v->stream));
        v->deferred = (cflags & VT_SRC_DEFERRED) != 0;
        v->lo_interior = (cflags & VT_SLAB_LO_INTERIOR) != 0;
        *out = v;
        return 0;
    }
    if (pad > 0) {
        // dense copy of the caller's samples on the device, then one pass writes the mirrored, padded resident copy
        const size_t ubytes = (size_t)uD * uH * uW * sizeof(float);
        float* d_dense = nullptr;
        VT_HIPC(cached_malloc(dev, reinterpret_cast<void**>(&d_dense), ubytes));
        hipError_t ec;
        {
            PinnedScope pin((cflags & VT_SRC_DEVICE) ? nullptr : data, (cflags & VT_SRC_DEVICE) ? 0 : ubytes);
            ec = (cflags & VT_SRC_DEVICE) ? hipMemcpyAsync(d_dense, data, ubytes, kind, v->stream) : pin.copy(d_dense, data, ubytes, kind, v->stream);
            if (ec == hipSuccess) ec = launch_mirror_pad(d_dense, v->d_src, uD, uH, uW, pad, v->P, v->stream);
            if (ec == hipSuccess) ec = hipStreamSynchronize(v->stream);
        }
        cached_free(dev, d_dense, ubytes);
        if (ec != hipSuccess) return cleanup(fail((int)ec, "mirror padding: %s", hipGetErrorString(ec)));
        rc = finalize_resident(v, false);
        if (rc) return cleanup(rc);
        *out = v;
        return 0;
    }
    // only the pad columns need zeroing (a full memset would cost another 4 B/voxel of HBM writes)
    VT_HIPC(hipMemset2DAsync(v->d_src + W, (size_t)v->P * sizeof(float), 0, (size_t)(v->P - W) * sizeof(float), (size_t)D * H, v->stream));
    {
        PinnedScope pin((cflags & VT_SRC_DEVICE) ? nullptr : data, (cflags & VT_SRC_DEVICE) ? 0 : (size_t)D * H * W * sizeof(float));
        if (cflags & VT_SRC_DEVICE)
            VT_HIPC(hipMemcpy2DAsync(v->d_src, (size_t)v->P * sizeof(float), data, (size_t)W * sizeof(float),
                                     (size_t)W * sizeof(float), (size_t)D * H, kind, v->stream));
        else
            VT_HIPC(pin.copy2d(v->d_src, (size_t)v->P * sizeof(float), data, (size_t)W * sizeof(float), (size_t)D * H, true, v->stream));
        VT_HIPC(hipStreamSynchronize(v->stream));
    }

    rc = finalize_resident(v, (cflags & VT_SLAB_LO_INTERIOR) != 0);
    if (rc) return cleanup(rc);
#undef VT_HIPC
    *out = v;
    return 0;
}

// Pipelined one-shot.  The source is uploaded in chunks of planes into ONE resident buffer on a non-blocking
// copy stream; the transform is launched per output slab on the handle's stream as soon as the last chunk that slab taps
// has arrived (stream waits on the upload events); every finished slab is downloaded on a second non-blocking copy stream
// while later chunks are still going up.  The sequence is the one of tools/probes/pipeline_probe.hip (512^3: 12.3 ms
// against 19.8 ms sequential).  Axis-0-separable matrices stream (each output plane taps a window of source
// planes), only the plain resident layout (the secondary copies are built from a complete source).  filt_* interpolations:
// the X and Y passes of the prefilter are plane-local and run per uploaded chunk; the axis-0 pass already works in chunks
// of 64 / 128 planes with 16 planes of warm-up, so each of its chunks is launched as soon as those planes are there.  (A
// resident volume filters axes 0 and 1 with the block-form kernel since round 2: the pipeline's coefficients agree with a
// resident volume's to ~1e-9 relative, not bit for bit; tests/test_gpu_parity.py compares the two paths on filt_*.)  Returns 1
// when the call does not qualify.
// VT_PIPE_TRACE: host-side timeline and per-chunk event times of one pipelined one-shot call
void pipeline_trace(double t_begin, double t_created, double t_pinned, double t_uploads, double t_slabs, double t_done, int nch,
                    const std::vector<hipEvent_t>& ev_up, const std::vector<hipEvent_t>& ev_k, const std::vector<hipEvent_t>& ev_dn)
{
    std::fprintf(stderr, "[pipe] host: create %.2f  alloc+pin %.2f  enqueue uploads %.2f  enqueue slabs %.2f  drain %.2f  total %.2f ms\n",
                 t_created - t_begin, t_pinned - t_created, t_uploads - t_pinned, t_slabs - t_uploads, t_done - t_slabs, t_done - t_begin);
    for (int k = 0; k < nch; ++k) {
        float a = 0, b = 0, c = 0;
        hipEventElapsedTime(&a, ev_up[0], ev_up[(size_t)k]);
        hipEventElapsedTime(&b, ev_up[0], ev_k[(size_t)k]);
        hipEventElapsedTime(&c, ev_up[0], ev_dn[(size_t)k]);
        std::fprintf(stderr, "[pipe] chunk %2d: upload done %+7.2f  kernel done %+7.2f  download done %+7.2f ms\n", k, a, b, c);
    }
}

// may this call take the pipelined path?  (see oneshot_pipelined)
bool pipeline_eligible(const float* h_volume, int D, int H, int W, int interp, const float* m4x4, const float* h_out, int flags)
{
    const size_t n = (size_t)D * H * W;
    if ((flags & (VT_KEEP_OUTSIDE | VT_FORCE_DIRECT | VT_NO_ZSEP | VT_NO_MARCH)) || n * sizeof(float) < ((size_t)32 << 20) || D < 32)
        return false;
    {   // result written over the input (output=volume): slabs would be downloaded over planes still waiting to be uploaded
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(h_volume), b0 = reinterpret_cast<uintptr_t>(h_out);
        if (a0 < b0 + n * sizeof(float) && b0 < a0 + n * sizeof(float)) return false;
    }
    // prefilter passes as run_prefilter orders them for such a volume: X in place, Y into the partner buffer, Z back
    if (is_filtered(interp) && (W > 2048 || prefilter_axis_in_place_ok(1, D, H, W) || prefilter_axis_in_place_ok(0, D, H, W))) return false;
    double m[12];
    for (int i = 0; i < 12; ++i) m[i] = (double)m4x4[i];
    for (int i = 0; i < 12; ++i)
        if (!std::isfinite(m[i])) return false;
    // (axis-0-separable matrices stream: an output slab needs a window of source planes.  Any other matrix needs the whole source
    // before its first output voxel; the pipeline still hides the plane-local prefilter passes behind the uploads and the transform
    // behind the downloads -- the floor is the two PCIe transfers back to back.  [measured, tools/oneshot_time.py] 512^3 general
    // rotation: filt_bspline 21.06 -> 20.11 ms, linear 19.76 -> 20.00 (nothing to hide but the chunking's own cost); 250^3 no gain.)
    const bool separable = m[0] == 1.0 && m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[8] == 0.0;
    if (!separable && !(is_filtered(interp) && n * sizeof(float) >= ((size_t)256 << 20))) return false;
    return std::fabs(m[3]) < 1.0e9;
}

// Three streams created back to back: the runtime deals hardware queues to streams round-robin (4 queues), and a
// download that shares its hardware queue with the kernels is executed in that queue, in order, by a shader copy at half
// the PCIe rate ([measured] kernel j only started when download j-1 had finished, 27 GB/s) -- so the pipeline's kernels
// run on a stream of their own, created with the two copy streams, not on whichever stream the handle got.
bool pipeline_streams(int dev)
{
    hipStream_t& s_up = g_cache[dev].copy_up;
    hipStream_t& s_dn = g_cache[dev].copy_dn;
    hipStream_t& s_k = g_cache[dev].pipe_k;
    if (s_up && s_dn && s_k) return true;
    if (hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s_dn, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        s_up = s_dn = s_k = nullptr;         // (a leaked stream on this error path is harmless)
        return false;
    }
    return true;
}

int oneshot_pipelined(int dev, const float* h_volume, int D, int H, int W, int interp, const float* m4x4, float* h_out, int flags)
{
    const size_t n = (size_t)D * H * W;
    if (dev >= 64 || !pipeline_eligible(h_volume, D, H, W, interp, m4x4, h_out, flags)) return 1;
    const bool filt = is_filtered(interp);
    double m[16];
    for (int i = 0; i < 16; ++i) m[i] = (double)m4x4[i];
    std::unique_lock<std::mutex> pipe_lock(g_cache[dev].pipe_mu, std::try_to_lock);
    if (!pipe_lock.owns_lock()) return 1;          // another thread is in the pipeline on this device: plain sequence
    if (!pipeline_streams(dev)) return 1;
    hipStream_t s_up = g_cache[dev].copy_up, s_dn = g_cache[dev].copy_dn, s_k = g_cache[dev].pipe_k;

    static const bool trace = std::getenv("VT_PIPE_TRACE") != nullptr;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_ms();
    vt_volume_t* v = nullptr;
    int rc = create_common(dev, D, H, W, interp, nullptr, kSrcNone, 0, D, 0, D, &v);
    if (rc) return rc;
    const double t_created = now_ms();
    hipStream_t own_stream = v->stream;           // the slab launches go to the pipeline's kernel stream
    v->stream = s_k;
    float* d_out = nullptr;
    float* d_tmp = nullptr;                        // prefilter ping-pong partner (filt_* only)
    const size_t src_bytes = (size_t)D * H * v->P * sizeof(float);
    int nch = D >= 256 ? 16 : 8;
    if (const char* e = std::getenv("VT_PIPE_NCH")) nch = std::max(1, std::min(D / 2, std::atoi(e)));
    const int Dc = (D + nch - 1) / nch;
    std::vector<hipEvent_t> ev_up((size_t)nch, nullptr), ev_k((size_t)nch, nullptr), ev_dn((size_t)nch, nullptr);
    auto finish = [&](int code) {
        (void)hipStreamSynchronize(s_up);
        (void)hipStreamSynchronize(s_k);
        (void)hipStreamSynchronize(s_dn);
        v->stream = own_stream;
        for (hipEvent_t e : ev_up) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_k) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_dn) if (e) (void)hipEventDestroy(e);
        cached_free(dev, d_out, n * sizeof(float));
        cached_free(dev, d_tmp, src_bytes);
        vt_volume_destroy(v);
        return code;
    };
#define VT_HIPP(call)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return finish(fail((int)e_, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__)); \
    } while (0)
    VT_HIPP(cached_malloc(dev, reinterpret_cast<void**>(&d_out), n * sizeof(float)));
    if (filt) VT_HIPP(cached_malloc(dev, reinterpret_cast<void**>(&d_tmp), src_bytes));
    // dependency-only events (no timestamps) unless the timeline is being traced
    const unsigned evflags = trace ? hipEventDefault : hipEventDisableTiming;
    for (int k = 0; k < nch; ++k) {
        VT_HIPP(hipEventCreateWithFlags(&ev_up[(size_t)k], evflags)); VT_HIPP(hipEventCreateWithFlags(&ev_k[(size_t)k], evflags));
        if (trace) VT_HIPP(hipEventCreateWithFlags(&ev_dn[(size_t)k], evflags));
    }
    PinnedScope pin_in(h_volume, n * sizeof(float)), pin_out(h_out, n * sizeof(float));
    const double t_pinned = now_ms();

    // uploads: all queued at once, in plane order (events complete in order)
    for (int k = 0; k < nch; ++k) {
        const int z0 = k * Dc, z1 = std::min(D, z0 + Dc);
        if (z0 < z1)
            VT_HIPP(pin_in.copy2d(v->d_src + (size_t)z0 * H * v->P, (size_t)v->P * sizeof(float), h_volume + (size_t)z0 * H * W,
                                  (size_t)W * sizeof(float), (size_t)(z1 - z0) * H, true, s_up));
        VT_HIPP(hipEventRecord(ev_up[(size_t)k], s_up));
    }
    const double t_uploads = now_ms();
    const int halo = is_cubic(interp) ? 2 : 1;
    const bool separable = m[0] == 1.0 && m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[8] == 0.0;
    const int lflags = (flags | VT_OUT_DEVICE | VT_NO_ZPAIR | VT_NO_QUAD | VT_NO_RSWAP) & ~VT_KEEP_OUTSIDE;
    // Kernel stream, per uploaded chunk: (filt_*) X and Y passes of the prefilter on the chunk's planes (plane-local), then
    // every axis-0 chunk of the prefilter whose input planes (its own + warm-up) are there, then every output slab whose
    // source planes are final.  The axis-0 pass uses the chunked kernel (prefilter_chunk_size / prefilter_warmup describe ITS grid).
    const int zC = filt ? prefilter_chunk_size(D) : 1;
    const int nzc = filt ? (D + zC - 1) / zC : 0;
    int z_chunks_done = 0, next_slab = 0;
    const int nslabs = (D + Dc - 1) / Dc;
    for (int k = 0; k < nch; ++k) {
        const int z0 = k * Dc, z1 = std::min(D, z0 + Dc);
        if (z0 >= z1) break;
        VT_HIPP(hipStreamWaitEvent(s_k, ev_up[(size_t)k], 0));
        int final_planes = z1;                     // source planes [0, final_planes) hold what the transform samples
        if (filt) {
            const size_t off = (size_t)z0 * H * v->P;
            if (prefilter_xy_ok(z1 - z0, H, W, v->P, v->d_src + off, d_tmp + off)) {
                VT_HIPP(launch_prefilter_xy(v->d_src + off, d_tmp + off, z1 - z0, H, W, v->P, s_k));
            } else {
                VT_HIPP(launch_prefilter_axis(2, v->d_src + off, v->d_src + off, z1 - z0, H, W, v->P, false, s_k));
                VT_HIPP(launch_prefilter_axis(1, v->d_src + off, d_tmp + off, z1 - z0, H, W, v->P, false, s_k));
            }
            int c1 = z_chunks_done;
            while (c1 < nzc && std::min(D, (c1 + 1) * zC + prefilter_warmup()) <= z1) ++c1;
            if (c1 > z_chunks_done) {
                VT_HIPP(launch_prefilter_axis0_chunks(d_tmp, v->d_src, D, H, W, v->P, z_chunks_done, c1, s_k));
                z_chunks_done = c1;
            }
            final_planes = std::min(D, z_chunks_done * zC);
        }
        while (next_slab < nslabs) {
            const int d0 = next_slab * Dc, d1 = std::min(D, d0 + Dc);
            // output plane d of an axis-0-separable map taps source planes floor(d + tz) - halo + 1 ... floor(d + tz) + halo; any
            // other matrix (and the secondary copies its kernels may build) needs every plane
            const double hi = separable ? std::floor((double)(d1 - 1) + m[3]) + halo : (double)(D - 1);
            if (std::min(hi, (double)(D - 1)) >= (double)final_planes) break;          // wait for more planes
            v->out_plane0 = d0;
            v->oD = d1 - d0;
            rc = do_affine(v, m, d_out + (size_t)d0 * H * W, lflags);
            v->out_plane0 = 0;
            v->oD = D;
            if (rc) return finish(rc);
            VT_HIPP(hipEventRecord(ev_k[(size_t)next_slab], s_k));
            ++next_slab;
        }
    }
    if (next_slab < nslabs) return finish(fail(VT_EINVAL, "one-shot pipeline left %d slabs unlaunched", nslabs - next_slab));
    // downloads: each one is queued when its slab is done (host wait -- the calling thread has nothing else to do)
    for (int j = 0; j < nslabs; ++j) {
        const int d0 = j * Dc, d1 = std::min(D, d0 + Dc);
        VT_HIPP(hipEventSynchronize(ev_k[(size_t)j]));
        VT_HIPP(pin_out.copy(h_out + (size_t)d0 * H * W, d_out + (size_t)d0 * H * W, (size_t)(d1 - d0) * H * W * sizeof(float),
                             hipMemcpyDeviceToHost, s_dn));
        if (trace) VT_HIPP(hipEventRecord(ev_dn[(size_t)j], s_dn));
    }
    const double t_slabs = now_ms();
    VT_HIPP(hipStreamSynchronize(s_dn));
    if (trace)
        pipeline_trace(t_begin, t_created, t_pinned, t_uploads, t_slabs, now_ms(), nch, ev_up, ev_k, ev_dn);
#undef VT_HIPP
    return finish(0);
}

// n transforms of one resident volume in one call.  Small volumes (the launch-latency regime: template
// rotations, sub-tomogram alignment) take ONE launch of the batched direct kernel, 65535 matrices at a time; larger
// ones are queued back to back on the handle's stream without returning to the caller in between.
int do_affine_batch(vt_volume* v, int n, const double* m4x4s, float* out, int flags)
{
    if (!v || !m4x4s || !out) return fail(VT_EINVAL, "NULL argument");
    if (v->deferred) return fail(VT_EINVAL, "handle has not been finalized (vt_volume_finalize)");
    if (n <= 0) return fail(VT_EINVAL, "batch size %d", n);
    int rc = use_device(v->dev);
    if (rc) return rc;
    const size_t n_out = (size_t)v->oD * v->oH * v->oW;
    const bool host_out = !(flags & VT_OUT_DEVICE);
    const bool small = n_out <= (size_t)96 * 96 * 96 && !(flags & VT_FORCE_TILED);
    if (!small) {
        for (int i = 0; i < n; ++i) {
            rc = do_affine(v, m4x4s + 16 * (size_t)i, out + (size_t)i * n_out, flags);
            if (rc) return rc;
        }
        return 0;
    }
    for (size_t i = 0; i < (size_t)n * 16; ++i)
        if (!std::isfinite(m4x4s[i])) return fail(VT_EINVAL, "matrix %zu entry %zu is not finite", i / 16, i % 16);
    // fold the output-plane / slab offsets exactly as do_affine does
    VT_HIP(hipStreamSynchronize(v->stream));          // the previous batch may still be reading the staging vector
    std::vector<double>& ms = v->h_batch_m;
    ms.resize((size_t)n * 12);
    for (int i = 0; i < n; ++i) {
        const double* a = m4x4s + 16 * (size_t)i;
        double* m = ms.data() + 12 * (size_t)i;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 4; ++c) m[4 * r + c] = a[4 * r + c];
            m[4 * r + 3] = std::fma(a[4 * r], (double)v->out_plane0, a[4 * r + 3]) + (double)v->edge_pad;
        }
        m[3] -= (double)v->plane0;
    }
    AffineParams p;
    std::memset(&p, 0, sizeof(p));
    TilePlan plan;
    plan_launch(v, ms.data(), (flags & VT_KEEP_OUTSIDE) | VT_FORCE_DIRECT, &p, &plan);   // dims, valid interval, flags
    if (v->batch_m_cap < ms.size()) {
        if (v->d_batch_m) { VT_HIP(hipFree(v->d_batch_m)); v->d_batch_m = nullptr; v->batch_m_cap = 0; }
        VT_HIP(hipMalloc(reinterpret_cast<void**>(&v->d_batch_m), ms.size() * sizeof(double)));
        v->batch_m_cap = ms.size();
    }
    // (pageable host memory never travels in pieces the runtime would pin in place: PinnedScope, rule 1)
    VT_HIP(PinnedScope::sliced(reinterpret_cast<char*>(v->d_batch_m), reinterpret_cast<const char*>(ms.data()), ms.size() * sizeof(double),
                               hipMemcpyHostToDevice, v->stream));
    float* d_out = out;
    const size_t total = n_out * (size_t)n;
    PinnedScope pin(host_out ? out : nullptr, host_out ? total * sizeof(float) : 0);
    if (host_out) {
        if (v->scratch_elems < total) {
            if (v->d_scratch_out) { cached_free(v->dev, v->d_scratch_out, v->scratch_elems * sizeof(float)); v->d_scratch_out = nullptr; v->scratch_elems = 0; }
            VT_HIP(cached_malloc(v->dev, reinterpret_cast<void**>(&v->d_scratch_out), total * sizeof(float)));
            v->scratch_elems = total;
        }
        d_out = v->d_scratch_out;
        if (flags & VT_KEEP_OUTSIDE) VT_HIP(pin.copy(d_out, out, total * sizeof(float), hipMemcpyHostToDevice, v->stream));
    }
    for (int first = 0; first < n; first += 65535) {
        const int cnt = std::min(65535, n - first);
        VT_HIP(launch_affine_direct_batch(v->interp, v->d_src, d_out + (size_t)first * n_out, v->d_batch_m + 12 * (size_t)first,
                                          cnt, p, v->stream));
    }
    v->last_kernel = 1;
    v->last_tile[0] = v->last_tile[1] = v->last_tile[2] = 0;
    v->last_lds[0] = v->last_lds[1] = v->last_lds[2] = 0;
    v->last_lds_bytes = 0; v->last_grid = (int)((n_out + 255) / 256);
    if (host_out) {
        VT_HIP(pin.copy(out, d_out, total * sizeof(float), hipMemcpyDeviceToHost, v->stream));
        VT_HIP(hipStreamSynchronize(v->stream));
    }
    return 0;
}

// Axis-0 projection: out[h, w] = sum_d affine(m)[d, h, w]  (see vt_kernels_project.hip)
int do_project(vt_volume* v, const double m4x4[16], float* out, int flags)
{
    if (!v || !m4x4 || !out) return fail(VT_EINVAL, "NULL argument");
    if (v->deferred) return fail(VT_EINVAL, "handle has not been finalized (vt_volume_finalize)");
    int rc = use_device(v->dev);
    if (rc) return rc;
    for (int i = 0; i < 12; ++i)
        if (!std::isfinite(m4x4[i])) return fail(VT_EINVAL, "matrix entry %d is not finite", i);
    // fold the output-plane / slab offsets exactly as do_affine does
    double m[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 4; ++c) m[4 * r + c] = m4x4[4 * r + c];
        m[4 * r + 3] = std::fma(m4x4[4 * r], (double)v->out_plane0, m4x4[4 * r + 3]);
    }
    m[3] -= (double)v->plane0;
    const bool host_out = !(flags & VT_OUT_DEVICE);
    const size_t n2 = (size_t)v->oH * v->oW;
    // (VT_EDGE_SCIPY handles: the fused plane sum assumes the texture contract's skirt; they transform, then sum)
    const bool zsep = v->edge_pad == 0 && !(flags & VT_NO_ZSEP) && m[0] == 1.0 && m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[8] == 0.0 &&
                      std::fabs(m[3]) < 1.0e9;
    if (zsep) {
        if (!v->proj) {
            const int kind = !is_cubic(v->interp) ? VT_LINEAR
                             : ((v->interp == VT_BSPLINE_SIMPLE || v->interp == VT_FILT_BSPLINE_SIMPLE) ? VT_BSPLINE_SIMPLE : VT_BSPLINE);
            vt_volume* h = nullptr;
            rc = create_common(v->dev, 3, v->H, v->W, kind, nullptr, kSrcNone, 0, 3, 1, 1, &h);
            if (rc) return rc;
            recycle_stream(h->dev, h->stream);   // the helper runs on this handle's stream (ordered after the sum)
            h->stream = v->stream;
            h->owns_stream = false;
            v->proj = h;
        }
        vt_volume* h = v->proj;
        h->oD = 1; h->oH = v->oH; h->oW = v->oW;
        ProjectParams q;
        std::memset(&q, 0, sizeof(q));
        q.D = v->D; q.H = v->H;
        q.vec = 4; q.nxv = (v->P - 4) / 4;
        q.src_pitch = v->P; q.dst_pitch = h->P; q.dst_plane = (int64_t)h->H * h->P; q.copies = 3;
        q.uniform = 0;
        const double fl = std::floor(m[3]);
        q.zoff = (int)fl;
        const float fz = (float)(m[3] - fl);
        if (!is_cubic(v->interp)) {
            q.halo = 0; q.ntap = 2; q.wz[0] = 1.0f - fz; q.wz[1] = fz;
        } else {
            q.halo = 1; q.ntap = 4;
            if (v->interp == VT_BSPLINE_SIMPLE || v->interp == VT_FILT_BSPLINE_SIMPLE) cubic_weights<true>(fz, q.wz);
            else cubic_weights<false>(fz, q.wz);
        }
        // valid output planes: 0 <= d < oD and the skirt rule vlo <= d + m[3] < vhi on the resident coordinates
        const double vlo = -0.5 - (double)v->plane0, vhi = (double)v->gD - 0.5 - (double)v->plane0;
        const double dlo = std::max(0.0, std::ceil(vlo - m[3])), dhi = std::min((double)v->oD - 1.0, std::ceil(vhi - m[3]) - 1.0);
        q.dlo = (int)std::max(-1.0e9, std::min(1.0e9, dlo));
        q.dhi = (int)std::max(-1.0e9, std::min(1.0e9, dhi));
        // The weighted plane sum depends on the axis-0 part of the map alone (offset m[3], output depth, slab offsets folded into
        // m[3]): a tilt series about the projection axis (examples/projections.py:20-26) asks for the same sum at every angle.
        // The helper keeps it; only the 2-D interpolation below runs again (512^3: 0.112 ms -> 0.01 ms per projection).
        if (v->tune.no_proj_cache || !(v->proj_sum_valid && v->proj_sum_m3 == m[3] && v->proj_sum_oD == v->oD && v->proj_sum_oplane0 == v->out_plane0)) {
            v->proj_sum_valid = false;
            VT_HIP(launch_plane_sum(v->d_src, h->d_src, q, v->stream));
            v->proj_sum_valid = true; v->proj_sum_m3 = m[3]; v->proj_sum_oD = v->oD; v->proj_sum_oplane0 = v->out_plane0;
        }
        double m2[16] = {1, 0, 0, 0, 0, m[5], m[6], m[7], 0, m[9], m[10], m[11], 0, 0, 0, 1};
        // the helper's planes change with every call: no cached pair copy, and a 2-D image does not need LDS staging
        rc = do_affine(h, m2, out, (flags & VT_OUT_DEVICE) | VT_FORCE_DIRECT);
        v->last_kernel = 7;
        v->last_tile[0] = v->last_tile[1] = v->last_tile[2] = 0;
        v->last_lds[0] = v->last_lds[1] = v->last_lds[2] = 0;
        v->last_lds_bytes = 0; v->last_grid = 0;
        return rc;
    }

    // general matrices: transform into scratch, then sum the planes
    const size_t n_out = (size_t)v->oD * n2;
    if (v->proj_tmp_elems < n_out) {
        if (v->d_proj_tmp) { VT_HIP(hipFree(v->d_proj_tmp)); v->d_proj_tmp = nullptr; v->proj_tmp_elems = 0; }
        VT_HIP(hipMalloc(reinterpret_cast<void**>(&v->d_proj_tmp), n_out * sizeof(float)));
        v->proj_tmp_elems = n_out;
    }
    rc = do_affine(v, m4x4, v->d_proj_tmp, (flags & ~VT_KEEP_OUTSIDE) | VT_OUT_DEVICE);
    if (rc) return rc;
    float* d_out = out;
    if (host_out) {
        if (v->scratch_elems < n2) {
            if (v->d_scratch_out) { cached_free(v->dev, v->d_scratch_out, v->scratch_elems * sizeof(float)); v->d_scratch_out = nullptr; v->scratch_elems = 0; }
            VT_HIP(cached_malloc(v->dev, reinterpret_cast<void**>(&v->d_scratch_out), n2 * sizeof(float)));
            v->scratch_elems = n2;
        }
        d_out = v->d_scratch_out;
    }
    ProjectParams q;
    std::memset(&q, 0, sizeof(q));
    q.D = v->oD; q.H = v->oH;
    q.vec = (v->oW % 4 == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0) ? 4 : 1;
    q.nxv = v->oW / q.vec;
    q.src_pitch = v->oW; q.dst_pitch = v->oW; q.dst_plane = 0; q.copies = 1; q.uniform = 1;
    VT_HIP(launch_plane_sum(v->d_proj_tmp, d_out, q, v->stream));
    if (host_out) {
        PinnedScope pin(out, n2 * sizeof(float));
        VT_HIP(pin.copy(out, d_out, n2 * sizeof(float), hipMemcpyDeviceToHost, v->stream));
        VT_HIP(hipStreamSynchronize(v->stream));
    }
    return 0;
}

}  // namespace

extern "C" {

const char* vt_last_error(void) { return g_err; }
const char* vt_version(void) { return "voltools_amd 0.1.0 (gfx950)"; }

int vt_device_count(int* count)
{
    if (!count) return fail(VT_EINVAL, "NULL count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess && n > 0) ? n : 0;     // "no device" is an answer, not an error
    return 0;
}

int vt_device_name(int dev, char* buf, int buflen)
{
    if (!buf || buflen <= 0) return fail(VT_EINVAL, "bad buffer");
    int rc = use_device(dev);
    if (rc) return rc;
    hipDeviceProp_t prop;
    VT_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, (size_t)buflen, "%s (%s)", prop.name, prop.gcnArchName);
    return 0;
}

int vt_device_props(int dev, int* cu_count, int* lds_bytes_per_block, uint64_t* hbm_bytes)
{
    int rc = use_device(dev);
    if (rc) return rc;
    hipDeviceProp_t prop;
    VT_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes_per_block) *lds_bytes_per_block = (int)prop.sharedMemPerBlock;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return 0;
}

int vt_device_synchronize(int dev)
{
    int rc = use_device(dev);
    if (rc) return rc;
    VT_HIP(hipDeviceSynchronize());
    return 0;
}

int vt_malloc(int dev, size_t bytes, void** dptr)
{
    if (!dptr || bytes == 0) return fail(VT_EINVAL, "bad arguments");
    int rc = use_device(dev);
    if (rc) return rc;
    VT_HIP(hipMalloc(dptr, bytes));
    return 0;
}

int vt_free(int dev, void* dptr)
{
    if (!dptr) return 0;
    int rc = use_device(dev);
    if (rc) return rc;
    VT_HIP(hipFree(dptr));
    return 0;
}

int vt_memset_zero(int dev, void* dptr, size_t bytes)
{
    if (!dptr) return fail(VT_EINVAL, "NULL pointer");
    int rc = use_device(dev);
    if (rc) return rc;
    VT_HIP(hipMemset(dptr, 0, bytes));
    return 0;
}

int vt_memcpy_h2d(int dev, void* dptr, const void* hptr, size_t bytes)
{
    if (!dptr || !hptr) return fail(VT_EINVAL, "NULL pointer");
    int rc = use_device(dev);
    if (rc) return rc;
    {
        PinnedScope pin(hptr, bytes);
        VT_HIP(pin.copy(dptr, hptr, bytes, hipMemcpyHostToDevice, nullptr));
        VT_HIP(hipStreamSynchronize(nullptr));
    }
    return 0;
}

int vt_memcpy_d2h(int dev, void* hptr, const void* dptr, size_t bytes)
{
    if (!dptr || !hptr) return fail(VT_EINVAL, "NULL pointer");
    int rc = use_device(dev);
    if (rc) return rc;
    {
        PinnedScope pin(hptr, bytes);
        VT_HIP(pin.copy(hptr, dptr, bytes, hipMemcpyDeviceToHost, nullptr));
        VT_HIP(hipStreamSynchronize(nullptr));
    }
    return 0;
}

int vt_memcpy_d2d(int dev, void* dst, const void* src, size_t bytes)
{
    if (!dst || !src) return fail(VT_EINVAL, "NULL pointer");
    int rc = use_device(dev);
    if (rc) return rc;
    VT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    return 0;
}

int vt_host_register(int dev, void* ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(VT_EINVAL, "NULL pointer or empty range");
    int rc = use_device(dev);
    if (rc) return rc;
    VT_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    PinnedScope::say("[vt pin] pool buffer [%p, %p) registered\n", ptr, (void*)((char*)ptr + bytes));
    return 0;
}

int vt_device_trim(int dev)
{
    if (dev < 0 || dev >= 64) return fail(VT_ENODEV, "device index %d", dev);
    int rc = use_device(dev);
    if (rc) return rc;
    std::vector<std::pair<size_t, void*>> drop, drop_small;
    {
        std::lock_guard<std::mutex> lk(g_cache[dev].mu);
        drop.swap(g_cache[dev].big);
        g_cache[dev].big_bytes = 0;
        drop_small.swap(g_cache[dev].bufs);
        g_cache[dev].buf_bytes = 0;
    }
    for (auto& d : drop) (void)hipFree(d.second);
    for (auto& d : drop_small) (void)hipFree(d.second);
    return 0;
}

int vt_host_unregister(int dev, void* ptr)
{
    if (!ptr) return 0;
    int rc = use_device(dev);
    if (rc) return rc;
    VT_HIP(hipHostUnregister(ptr));
    PinnedScope::say("[vt pin] pool buffer %p released\n", ptr);
    return 0;
}

int vt_volume_create(int dev, int depth, int height, int width, int interp, const float* data,
                     int create_flags, vt_volume_t** out)
{
    return create_common(dev, depth, height, width, interp, data, create_flags & (VT_SRC_DEVICE | VT_EDGE_SCIPY),
                         0, depth, 0, depth, out);
}

int vt_volume_create_slab(int dev, int local_depth, int height, int width, int interp, const float* data,
                          int create_flags, int64_t plane0, int64_t global_depth, int64_t out_plane0,
                          int out_depth, vt_volume_t** out)
{
    return create_common(dev, local_depth, height, width, interp, data, create_flags,
                         plane0, global_depth, out_plane0, out_depth, out);
}

int vt_volume_upload_planes(vt_volume_t* v, int first_plane, int nplanes, const float* data, int flags)
{
    if (!v || !data) return fail(VT_EINVAL, "NULL argument");
    if (!v->deferred) return fail(VT_EINVAL, "planes can only be uploaded into a handle created with VT_SRC_DEFERRED, before vt_volume_finalize");
    if (first_plane < 0 || nplanes <= 0 || (int64_t)first_plane + nplanes > v->D)
        return fail(VT_EINVAL, "planes [%d, %d) outside the resident window of %d planes", first_plane, first_plane + nplanes, v->D);
    int rc = use_device(v->dev);
    if (rc) return rc;
    const bool src_dev = (flags & VT_SRC_DEVICE) != 0;
    const size_t rows = (size_t)nplanes * v->H;
    PinnedScope pin(src_dev ? nullptr : data, src_dev ? 0 : rows * v->W * sizeof(float));
    if (src_dev)
        VT_HIP(hipMemcpy2DAsync(v->d_src + (size_t)first_plane * v->H * v->P, (size_t)v->P * sizeof(float), data, (size_t)v->W * sizeof(float),
                                (size_t)v->W * sizeof(float), rows, hipMemcpyDeviceToDevice, v->stream));
    else
        VT_HIP(pin.copy2d(v->d_src + (size_t)first_plane * v->H * v->P, (size_t)v->P * sizeof(float), data, (size_t)v->W * sizeof(float), rows, true, v->stream));
    VT_HIP(hipStreamSynchronize(v->stream));
    return 0;
}

int vt_volume_finalize(vt_volume_t* v)
{
    if (!v) return fail(VT_EINVAL, "NULL handle");
    if (!v->deferred) return 0;
    int rc = use_device(v->dev);
    if (rc) return rc;
    return finalize_resident(v, v->lo_interior);
}

int vt_volume_destroy(vt_volume_t* v)
{
    if (!v) return 0;
    hipSetDevice(v->dev);
    if (v->stream) hipStreamSynchronize(v->stream);
    if (v->d_src) cached_free(v->dev, v->d_src, v->src_bytes);
    if (v->d_queue) hipFree(v->d_queue);
#ifdef VT_LEGACY
    if (v->d_src_zp) hipFree(v->d_src_zp);
    if (v->d_src_t_zp) hipFree(v->d_src_t_zp);
    if (v->d_src_r_zp) hipFree(v->d_src_r_zp);
    if (v->d_src_x_zp) hipFree(v->d_src_x_zp);
#endif
    if (v->d_src_t) hipFree(v->d_src_t);
    if (v->d_src_x) hipFree(v->d_src_x);
    if (v->d_src_r) hipFree(v->d_src_r);
    if (v->d_src_q) hipFree(v->d_src_q);
    if (v->d_src_t_q) hipFree(v->d_src_t_q);
    if (v->d_src_r_q) hipFree(v->d_src_r_q);
    if (v->d_src_x_q) hipFree(v->d_src_x_q);
    for (int i = 0; i < 4; ++i) if (v->d_src_qe[i]) hipFree(v->d_src_qe[i]);
    if (v->d_src_xe) hipFree(v->d_src_xe);
    if (v->d_tmp_x) hipFree(v->d_tmp_x);
    if (v->spare) hipFree(v->spare);
    if (v->d_scratch_out) cached_free(v->dev, v->d_scratch_out, v->scratch_elems * sizeof(float));
    if (v->d_proj_tmp) hipFree(v->d_proj_tmp);
    if (v->d_batch_m) hipFree(v->d_batch_m);
    if (v->proj) { vt_volume_destroy(v->proj); v->proj = nullptr; }
    if (v->ev0) recycle_event(v->dev, v->ev0);
    if (v->ev1) recycle_event(v->dev, v->ev1);
    if (v->evc0) (void)hipEventDestroy(v->evc0);
    if (v->evc1) (void)hipEventDestroy(v->evc1);
    if (v->stream && v->owns_stream) recycle_stream(v->dev, v->stream);
    delete v;
    return 0;
}

// Free every resident copy a handle has built lazily besides its plain one (exchanged orientations, plane-quad and z-convolved
// plane-quad forms, the exchanged-result buffer).  They are rebuilt by the first call that needs them; results do not change.
int vt_volume_release_copies(vt_volume_t* v, uint64_t* freed_bytes)
{
    if (!v) return fail(VT_EINVAL, "NULL argument");
    int rc = use_device(v->dev);
    if (rc) return rc;
    vt_volume_info_t before, after;
    if ((rc = vt_volume_info(v, &before))) return rc;
    VT_HIP(hipStreamSynchronize(v->stream));      // no launch may still be reading what is freed
    float** const slots[] = {&v->d_src_t, &v->d_src_x, &v->d_src_r, &v->d_src_q, &v->d_src_t_q, &v->d_src_r_q, &v->d_src_x_q,
                             &v->d_src_qe[0], &v->d_src_qe[1], &v->d_src_qe[2], &v->d_src_qe[3], &v->d_tmp_x, &v->d_src_xe,
#ifdef VT_LEGACY
                             &v->d_src_zp, &v->d_src_t_zp, &v->d_src_r_zp, &v->d_src_x_zp,
#endif
    };
    for (float** s : slots)
        if (*s) { VT_HIP(hipFree(*s)); *s = nullptr; }
    drop_spare(v);
    v->tmp_x_elems = 0;
    for (int i = 0; i < 4; ++i) { v->quad_bytes[i] = 0; v->quade_bytes[i] = 0; }
    for (int i = 0; i < 8; ++i) v->copy_retry_in[i] = 0;      // memory was just returned: a copy that did not fit may fit now
    v->xe_retry_in = 0;
    if ((rc = vt_volume_info(v, &after))) return rc;
    if (freed_bytes) *freed_bytes = before.resident_bytes - after.resident_bytes;
    return 0;
}

int vt_has_legacy_kernels(void)
{
#ifdef VT_LEGACY
    return 1;
#else
    return 0;
#endif
}

int vt_volume_info(const vt_volume_t* v, vt_volume_info_t* info)
{
    if (!v || !info) return fail(VT_EINVAL, "NULL argument");
    info->device = v->dev; info->interp = v->interp;
    // the caller's dims: a VT_EDGE_SCIPY handle keeps edge_pad mirrored voxels on every side of the resident copy (v->D/H/W)
    info->depth = v->D - 2 * v->edge_pad; info->height = v->H - 2 * v->edge_pad; info->width = v->W - 2 * v->edge_pad;
    info->out_depth = v->oD; info->out_height = v->oH; info->out_width = v->oW;
    info->last_kernel = v->last_kernel;
    for (int i = 0; i < 3; ++i) { info->last_tile[i] = v->last_tile[i]; info->last_lds_dims[i] = v->last_lds[i]; }
    info->last_lds_bytes = v->last_lds_bytes; info->last_grid = v->last_grid;
    info->prefilter_ms = v->prefilter_ms;
    const uint64_t plain = (uint64_t)v->D * v->H * v->P * sizeof(float);
    info->resident_bytes = plain + (v->d_src_t ? plain : 0) + (v->d_src_xe ? plain : 0) +
                           (v->proj ? (uint64_t)3 * v->proj->H * v->proj->P * sizeof(float) : 0) +
                           (v->d_src_r ? (uint64_t)v->D * v->W * v->Pr * sizeof(float) : 0) +
                           (v->d_src_x ? (uint64_t)v->W * v->H * v->Px * sizeof(float) : 0) +
                           (v->d_tmp_x ? (uint64_t)v->tmp_x_elems * sizeof(float) : 0) +
                           v->quad_bytes[0] + v->quad_bytes[1] + v->quad_bytes[2] + v->quad_bytes[3] +
                           v->quade_bytes[0] + v->quade_bytes[1] + v->quade_bytes[2] + v->quade_bytes[3];
#ifdef VT_LEGACY
    info->resident_bytes += (v->d_src_zp ? (uint64_t)((v->D + 1) / 2) * v->H * v->P2 * sizeof(float) : 0) +
                            (v->d_src_t_zp ? (uint64_t)((v->H + 1) / 2) * v->D * v->P2 * sizeof(float) : 0) +
                            (v->d_src_r_zp ? (uint64_t)((v->D + 1) / 2) * v->W * v->P2 * sizeof(float) : 0) +
                            (v->d_src_x_zp ? (uint64_t)((v->W + 1) / 2) * v->H * v->P2 * sizeof(float) : 0);
#endif
    info->copies_ms = v->copies_ms;
    info->copies_built = v->copies_built;
    info->copies_evicted = v->copies_evicted;
    info->max_resident_bytes = v->max_resident;
    return 0;
}

// Resident-memory budget of a handle (round 5; no reference counterpart: the reference keeps one CUDA array per StaticVolume,
// volume.py:37-45).  `bytes` counts the plain resident copy and every lazily built one; 0 = no limit (the default unless
// VT_MAX_RESIDENT_GB is set).  A copy that would take the handle over its budget is built only after the least recently used lazy copies
// have been released; one that cannot fit at all is not built, and the call runs on the kernel family that samples the plain layout.
// A budget below the handle's present footprint releases copies at once.  Results never depend on the budget.
int vt_volume_set_max_resident(vt_volume_t* v, uint64_t bytes)
{
    if (!v) return fail(VT_EINVAL, "NULL argument");
    int rc = use_device(v->dev);
    if (rc) return rc;
    v->max_resident = bytes;
    if (bytes)
        while (resident_now(v) > bytes) {
            if (v->spare) { drop_spare(v); continue; }
            if (!evict_lru(v, -1, -1)) break;         // (the plain copy alone may exceed a tiny budget: it stays)
        }
    for (int i = 0; i < 8; ++i) v->copy_retry_in[i] = 0;
    v->xe_retry_in = 0;
    return 0;
}

int vt_volume_stream(const vt_volume_t* v, void** hip_stream)
{
    if (!v || !hip_stream) return fail(VT_EINVAL, "NULL argument");
    *hip_stream = reinterpret_cast<void*>(v->stream);
    return 0;
}

int vt_volume_sync(vt_volume_t* v)
{
    if (!v) return fail(VT_EINVAL, "NULL handle");
    int rc = use_device(v->dev);
    if (rc) return rc;
    VT_HIP(hipStreamSynchronize(v->stream));
    return 0;
}

int vt_volume_set_output_shape(vt_volume_t* v, int od, int oh, int ow)
{
    if (!v) return fail(VT_EINVAL, "NULL handle");
    if (od <= 0 || oh <= 0 || ow <= 0) return fail(VT_EINVAL, "non-positive output dims");
    if ((int64_t)od * oh > 0x7fffffffLL || (int64_t)oh * ow > 0x7fffffffLL) return fail(VT_EUNSUPPORTED, "output too large");
    v->oD = od; v->oH = oh; v->oW = ow;
    return 0;
}

int vt_volume_affine(vt_volume_t* v, const float* m4x4, float* out, int flags)
{
    if (!m4x4) return fail(VT_EINVAL, "NULL matrix");
    double m[16];
    for (int i = 0; i < 16; ++i) m[i] = (double)m4x4[i];
    return do_affine(v, m, out, flags);
}

int vt_volume_affine_f64(vt_volume_t* v, const double* m4x4, float* out, int flags)
{
    return do_affine(v, m4x4, out, flags);
}

int vt_volume_affine_batch(vt_volume_t* v, int n, const float* m4x4s, float* out, int flags)
{
    if (!m4x4s || n <= 0) return fail(VT_EINVAL, "NULL matrices or empty batch");
    std::vector<double> m((size_t)n * 16);
    for (size_t i = 0; i < m.size(); ++i) m[i] = (double)m4x4s[i];
    return do_affine_batch(v, n, m.data(), out, flags);
}

int vt_volume_project(vt_volume_t* v, const float* m4x4, float* out_hw, int flags)
{
    if (!m4x4) return fail(VT_EINVAL, "NULL matrix");
    double m[16];
    for (int i = 0; i < 16; ++i) m[i] = (double)m4x4[i];
    return do_project(v, m, out_hw, flags);
}

int vt_volume_project_f64(vt_volume_t* v, const double* m4x4, float* out_hw, int flags)
{
    return do_project(v, m4x4, out_hw, flags);
}

int vt_timer_start(vt_volume_t* v)
{
    if (!v) return fail(VT_EINVAL, "NULL handle");
    int rc = use_device(v->dev);
    if (rc) return rc;
    VT_HIP(hipEventRecord(v->ev0, v->stream));
    return 0;
}

int vt_timer_stop(vt_volume_t* v, float* ms)
{
    if (!v || !ms) return fail(VT_EINVAL, "NULL argument");
    int rc = use_device(v->dev);
    if (rc) return rc;
    VT_HIP(hipEventRecord(v->ev1, v->stream));
    VT_HIP(hipEventSynchronize(v->ev1));
    VT_HIP(hipEventElapsedTime(ms, v->ev0, v->ev1));
    return 0;
}

int vt_prefilter_inplace(int dev, float* d_volume, int D, int H, int W)
{
    if (!d_volume) return fail(VT_EINVAL, "NULL volume");
    if (D <= 0 || H <= 0 || W <= 0) return fail(VT_EINVAL, "non-positive dims");
    if ((int64_t)D * H > 0x7fffffffLL || (int64_t)H * W > 0x7fffffffLL) return fail(VT_EUNSUPPORTED, "volume too large");
    int rc = init_device(dev);
    if (rc) return rc;
    const size_t bytes = (size_t)D * H * W * sizeof(float);
    float* d_tmp = nullptr;
    VT_HIP(hipMalloc(reinterpret_cast<void**>(&d_tmp), bytes));
    float* res = nullptr;
    rc = run_prefilter(d_volume, d_tmp, D, H, W, W, false, nullptr, &res);
    if (!rc && res != d_volume) {
        hipError_t e = hipMemcpyAsync(d_volume, res, bytes, hipMemcpyDeviceToDevice, nullptr);
        if (e != hipSuccess) rc = fail((int)e, "copy back: %s", hipGetErrorString(e));
    }
    hipError_t es = hipStreamSynchronize(nullptr);
    hipFree(d_tmp);
    if (!rc && es != hipSuccess) rc = fail((int)es, "prefilter: %s", hipGetErrorString(es));
    return rc;
}

// One-shot transform of a host volume into a host buffer (transforms.py:164-226): upload, prefilter, transform, download.
// PCIe-bound (512^3: 2 x 9.4 ms of copies around < 1.5 ms of kernels).  Where the output planes only need nearby source
// planes (axis-0-separable matrices, no prefilter) the call is pipelined so that both PCIe directions run at once
// (oneshot_pipelined below); everything else is the plain sequence.
int vt_affine_oneshot(int dev, const float* h_volume, int D, int H, int W, int interp, const float* m4x4,
                      float* h_out, int flags, float* elapsed_ms)
{
    if (!h_volume || !m4x4 || !h_out) return fail(VT_EINVAL, "NULL argument");
    int rc = init_device(dev);
    if (rc) return rc;
    hipEvent_t t0, t1;
    VT_HIP(hipEventCreate(&t0));
    VT_HIP(hipEventCreate(&t1));
    hipEventRecord(t0, nullptr);
    vt_volume_t* v = nullptr;
    static const bool no_pipe = std::getenv("VT_ONESHOT_SEQ") != nullptr;      // A/B switch: always the plain sequence
    const bool edge = (flags & VT_ONESHOT_EDGE_SCIPY) != 0;
    flags &= ~VT_ONESHOT_EDGE_SCIPY;
    rc = (no_pipe || edge) ? 1 : oneshot_pipelined(dev, h_volume, D, H, W, interp, m4x4, h_out, flags & ~VT_OUT_DEVICE);
    if (rc == 1) {
        rc = vt_volume_create(dev, D, H, W, interp, h_volume, edge ? VT_EDGE_SCIPY : 0, &v);
        if (!rc) rc = vt_volume_affine(v, m4x4, h_out, flags & ~VT_OUT_DEVICE);
    }
    hipEventRecord(t1, nullptr);
    hipEventSynchronize(t1);
    if (elapsed_ms) hipEventElapsedTime(elapsed_ms, t0, t1);
    hipEventDestroy(t0);
    hipEventDestroy(t1);
    vt_volume_destroy(v);
    return rc;
}

}  // extern "C"
