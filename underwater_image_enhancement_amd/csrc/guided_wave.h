// Shared pieces of the wavefront guided-filter kernels (k_guided_pipe.hip).
#pragma once
#include "common.h"
#include "devutil.h"

#include <type_traits>
#include <utility>

namespace uwie {
namespace {

constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52: fma(x, s, kMagic) has round-to-nearest(x*s) in its low word


typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct PipeConsts {
    double Ek;      // 255 * K^4 * eps
    double fxa;     // 2^Sa                     (FX32)
    double fxb;     // scale * 2^Sb             (FX32: b_fixed = lo32(fma(t3, fxb, magic_b)))
    double magic_b; // kMagic - round(b0 * 2^Sb)
    double kaI;     // mean_a * I = SA * kaI * g     (scale / 255 [/ 2^Sa])
    double kb, b0;  // mean_b = SB * kb + b0
};

__device__ __forceinline__ void pipe_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int pipe_reflect(int p, int len)
{
    if (p < 0) p = -p;
    if (p > len - 1) p = 2 * (len - 1) - p;
    return min(max(p, 0), len - 1);
}

__device__ __forceinline__ double pipe_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, y, 1.0), y, y);
}

// Raw buffer resource over one image plane (stride 0): loads take a per-lane byte offset (VGPR) plus a wave-uniform row
// offset (SGPR), so the row loop needs no vector address arithmetic; accesses at or beyond `bytes` are dropped by the
// hardware's range check, which is also how lanes without a valid output column skip their store (kNoStore).
constexpr uint32_t kNoStore = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pipe_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t pipe_opaque(uint32_t v)  // keeps a loop-invariant address in its register
{
    asm volatile("" : "+v"(v));
    return v;
}

template <int N>
using IC = std::integral_constant<int, N>;


struct SplitGeom {
    int H, W, y0, band;  // bands of `band` rows (a multiple of the ring period) from row y0, every strip
    int y_end;           // the last band runs to this row (shorter or longer than `band`; whole ring periods unless it is H)
    int xcd_fold;        // fold launch order so that an XCD's workgroups are neighbours (see k_guided_split)
    int nbands;          // bands per strip (the last one runs to y_end)
};

// Neighbouring strips share the cache lines their halos overlap in (a strip's rows start at arbitrary bytes).  Workgroups are
// dealt to the 8 XCDs round-robin in launch order, so launch order is folded: the workgroups one XCD receives form one
// contiguous run of (strip, band, image) triples, strips fastest, and neighbours run at the same time behind the same L2.
__device__ __forceinline__ int3 xcd_folded_block()
{
    const uint32_t gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    const uint32_t lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const uint32_t xcd = lin & 7u, idx = lin >> 3, q = n >> 3, r = n & 7u;
    const uint32_t log = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    return int3{(int)(log % gx), (int)((log / gx) % gy), (int)(log / (gx * gy))};
}

}  // namespace
}  // namespace uwie
