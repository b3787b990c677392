// What does v_ashr_pk_u8_i32 (new on gfx950) compute?  Round 3 found the compiler selecting it for two adjacent
// saturate_cast<uchar>(x >> 15) values and the bytes coming out wrong on the hardware ("A compiler finding").
// This program runs the instruction itself (inline assembly) and the C++ pattern the compiler turns into it on random operands
// and compares both with  sat_u8(x0 >> s) | sat_u8(x1 >> s) << 8.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ashr_pk profiles/microbench/ashr_pk.hip && /tmp/ashr_pk
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_asm(const int *x0, const int *x1, const int *sh, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d = 0xdeadbeefu;  // (the instruction writes 16 bits: what happens to the other 16?)
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "+v"(d) : "v"(x0[i]), "v"(x1[i]), "v"(sh[i]));
    out[i] = d;
}

__device__ __forceinline__ uint32_t sat_u8(int v) { return (uint32_t)min(max(v, 0), 255); }

// the pattern of round 3's RGB2LAB: shift, then clamp, two adjacent values packed
__global__ void k_cxx(const int *x0, const int *x1, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = sat_u8(x0[i] >> 15) | (sat_u8(x1[i] >> 15) << 8);
}

int main()
{
    const int n = 1 << 20;
    std::vector<int> x0(n), x1(n), sh(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int i = 0; i < n; ++i) {
        const int kind = i & 3;  // 0: the LAB range (around 2^23), 1: full range, 2: small, 3: negative-heavy
        auto gen = [&]() -> int {
            const uint32_t r = rnd();
            return kind == 0 ? (int)(r % (300u << 15)) - (20 << 15) : kind == 1 ? (int)r : kind == 2 ? (int)(r & 0xffff) - 0x4000 : -(int)(r & 0x7fffffff);
        };
        x0[i] = gen();
        x1[i] = gen();
        sh[i] = (i & 4) ? 15 : (int)(rnd() & 63);  // (shift operands above 31 too: which bits count?)
    }
    int *d0, *d1, *ds;
    uint32_t *da, *dc;
    CHECK(hipMalloc(&d0, n * 4)); CHECK(hipMalloc(&d1, n * 4)); CHECK(hipMalloc(&ds, n * 4));
    CHECK(hipMalloc(&da, n * 4)); CHECK(hipMalloc(&dc, n * 4));
    CHECK(hipMemcpy(d0, x0.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d1, x1.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(ds, sh.data(), n * 4, hipMemcpyHostToDevice));
    k_asm<<<n / 256, 256>>>(d0, d1, ds, da, n);
    k_cxx<<<n / 256, 256>>>(d0, d1, dc, n);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> a(n), c(n);
    CHECK(hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost));
    auto sat = [](long long v) { return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
    // hypotheses about the instruction: which bits of the shift operand count, and where the two bytes land
    struct Hyp { const char *name; int mask; bool swap; } hyps[] = {{"s & 31, x0 -> byte 0", 31, false}, {"s & 31, x0 -> byte 1", 31, true},
                                                                   {"s & 63 (arith., saturating at 31), x0 -> byte 0", 63, false}};
    for (const Hyp &h : hyps) {
        long bad = 0;
        for (int i = 0; i < n; ++i) {
            const int sv = sh[i] & h.mask, se = sv > 31 ? 31 : sv;
            const uint32_t b0 = sat((long long)x0[i] >> se), b1 = sat((long long)x1[i] >> se);
            const uint32_t want = h.swap ? (b1 | b0 << 8) : (b0 | b1 << 8);
            bad += (a[i] & 0xffffu) != want;
        }
        printf("inline asm, low 16 bits == sat_u8(x >> (%s)): %ld of %d differ\n", h.name, bad, n);
    }
    long keep = 0, zero = 0;
    for (int i = 0; i < n; ++i) {
        keep += (a[i] >> 16) == 0xdeadu;
        zero += (a[i] >> 16) == 0;
    }
    printf("inline asm, high 16 bits of the destination: kept in %ld, zero in %ld of %d\n", keep, zero, n);
    long badc = 0;
    int shown = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t want = sat((long long)x0[i] >> 15) | sat((long long)x1[i] >> 15) << 8;
        if (c[i] != want) {
            ++badc;
            if (shown++ < 6) printf("  C++ pattern: x0 = %d, x1 = %d: got 0x%08x, want 0x%08x\n", x0[i], x1[i], c[i], want);
        }
    }
    printf("C++ shift-then-clamp pattern (compiled by this toolchain): %ld of %d differ from sat_u8(x >> 15) pairs\n", badc, n);
    return 0;
}
