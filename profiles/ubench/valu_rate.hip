// Issue rate of a few integer VALU instructions on gfx950, one wave per SIMD (4 waves per block, one block per CU):
//   hipcc --offload-arch=gfx950 -O3 profiles/ubench/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
// Prints wave-cycles per instruction (s_memtime ticks at 100 MHz are converted with the measured shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t a0, uint32_t b0, int iters)
{
    uint32_t acc[8];
    uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
    for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) acc[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), acc[i], false);
                if (OP == 1) acc[i] = __umul24(a, b) + acc[i];
                if (OP == 2) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (OP == 3) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
                if (OP == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
                if (OP == 5) acc[i] = __builtin_amdgcn_udot4(a, b, acc[i], false);
                if (OP == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t *)&acc[i & 6]) : "v"(a), "v"(b) : "vcc");
            }
            if (OP <= 1 || OP == 5) a += acc[0] & 1;  // keep the compiler from folding the chain
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(const char *name, uint32_t *d)
{
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, 256>>>(d, 3, 5, 16);
    hipEventRecord(e0);
    k<OP><<<256, 256>>>(d, 3, 5, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 16 * 8;  // per wave; one wave per SIMD
    const double ns_per = ms * 1e6 / instr;
    printf("%-16s %8.3f ms  %6.2f ns/instr/wave  = %5.2f cycles at 2.4 GHz\n", name, ms, ns_per, ns_per * 2.4);
    return ns_per;
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, 256 * 256 * 4);
    run<4>("v_add_u32", d);
    run<1>("v_mad_u32_u24", d);
    run<2>("v_mad_u32_u16", d);
    run<0>("v_dot2_u32_u16", d);
    run<5>("v_dot4_u32_u8", d);
    run<3>("v_mul_lo_u32", d);
    run<6>("v_mad_u64_u32", d);
    return 0;
}
