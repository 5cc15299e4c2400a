// vt_march_common.h -- pieces shared by the marching kernels (vt_kernels_march.hip: plain / plane-pair layouts,
// vt_kernels_quad.hip: plane-quad layout): counted vector-memory waits, the packed row-span table of a tile's footprint.
#pragma once
#include "vt_internal.h"
#include "vt_device.h"

namespace vt {

// wait until at most n of this wave's vector-memory operations are outstanding (n is wave-uniform; the instruction
// needs an immediate)
__device__ __forceinline__ void wait_vmcnt_le(int n)
{
#define VT_WCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        VT_WCASE(0) VT_WCASE(1) VT_WCASE(2) VT_WCASE(3) VT_WCASE(4) VT_WCASE(5) VT_WCASE(6) VT_WCASE(7)
        VT_WCASE(8) VT_WCASE(9) VT_WCASE(10) VT_WCASE(11) VT_WCASE(12) VT_WCASE(13) VT_WCASE(14) VT_WCASE(15)
        VT_WCASE(16) VT_WCASE(17) VT_WCASE(18) VT_WCASE(19) VT_WCASE(20) VT_WCASE(21) VT_WCASE(22) VT_WCASE(23)
        VT_WCASE(24) VT_WCASE(25) VT_WCASE(26) VT_WCASE(27) VT_WCASE(28) VT_WCASE(29) VT_WCASE(30) VT_WCASE(31)
        VT_WCASE(32) VT_WCASE(33) VT_WCASE(34) VT_WCASE(35) VT_WCASE(36) VT_WCASE(37) VT_WCASE(38) VT_WCASE(39)
        VT_WCASE(40) VT_WCASE(41) VT_WCASE(42) VT_WCASE(43) VT_WCASE(44) VT_WCASE(45) VT_WCASE(46) VT_WCASE(47)
        VT_WCASE(48) VT_WCASE(49) VT_WCASE(50) VT_WCASE(51) VT_WCASE(52) VT_WCASE(53) VT_WCASE(54) VT_WCASE(55)
        default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    }
#undef VT_WCASE
}

constexpr int kRowsMax = 64;      // source rows a tile's footprint may span (host-checked)
constexpr int kMaxIt = 4;         // packed footprint <= 1024 vectors (16 KiB) per plane (host-checked via slot size)
constexpr int kTabInts = 2 * kRowsMax + 4;                 // x0[64] | first[64] | total | pad
constexpr int kVrowCap = 1024 * kMaxIt;                    // widest workgroup (1024 threads) x kMaxIt vectors
constexpr int kTabBytes = kTabInts * 4 + kVrowCap;         // + row index of every vector (host: march_table_bytes())

// Row spans of a tile's footprint, packed (both marching kernels).  Row `tid` of the box: which columns do the tile's
// pixels tap there (march_row_span)?  Spans are aligned to ALIGN positions (= one 16-byte vector) and laid back to back:
//   tab[0..63] = aligned span start x0, tab[64..127] = first vector of the row, tab[128] = vectors per plane,
//   vrow[v]    = row of vector v (every row's lane writes its own vectors' entries: no search afterwards).
// The table overlays the ring, which is not in use yet.  Ends with a barrier.
template <int TH, int TW, int HALO, int ALIGN>
__device__ __forceinline__ void build_span_table(int* tab, const AffineParams& p, int Ly, double by, double bx, int tid)
{
    unsigned char* vrow = reinterpret_cast<unsigned char*>(tab + kTabInts);
    if (tid < kRowsMax) {
        const int lane = tid;
        int mn, mx;
        const bool used = (tid < Ly) && march_row_span(p.m[5], p.m[6], p.m[9], p.m[10], p.ia1, p.ib1, by, bx, tid, TH, TW, HALO, &mn, &mx);
        const int x0 = used ? (mn & ~(ALIGN - 1)) : 0;
        const int nv = used ? ((mx - x0) / ALIGN + 1) : 0;
        int incl = nv;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) {
            const int up = __shfl_up(incl, s2);
            if (lane >= s2) incl += up;
        }
        const int first = incl - nv;
        tab[tid] = x0;
        tab[kRowsMax + tid] = first;
        if (tid == kRowsMax - 1) tab[2 * kRowsMax] = incl;
        const int last = min(first + nv, kVrowCap);            // larger footprints take the fallback path anyway
        for (int v = first; v < last; ++v) vrow[v] = (unsigned char)tid;
    }
    __syncthreads();
}

}  // namespace vt
