// Issue-rate microbenchmark for the instruction classes the guided filter is built from (gfx950).
// Reports cycles per wave-instruction per SIMD at 1, 2 and 4 waves per SIMD (8 independent chains per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 2048;

#define DEF_KERNEL(NAME, DECL, BODY, SINK)                                              \
__global__ void __launch_bounds__(256) NAME(unsigned long long *out, double seed) {     \
    DECL                                                                                \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
    for (int it = 0; it < ITERS; ++it) {                                                \
        BODY                                                                            \
    }                                                                                   \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
    SINK                                                                                \
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0; \
}

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// ---- f32
#define D_F32 float a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; float c=(float)seed*0.5f;
#define S_F32 if (a0+a1+a2+a3+a4+a5+a6+a7 == 12345.f) out[0] = 1;
#define B_ADDF32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_add_f32, D_F32, R8(B_ADDF32), S_F32)
#define B_FMAF32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_fma_f32, D_F32, R8(B_FMAF32), S_F32)
#define B_CVTU32F32(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a##i));
DEF_KERNEL(k_cvt_u32_f32, D_F32, R8(B_CVTU32F32), S_F32)
#define B_RCPF32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
DEF_KERNEL(k_rcp_f32, D_F32, R8(B_RCPF32), S_F32)

// ---- u32
#define D_U32 unsigned a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; unsigned c=(unsigned)seed*3u+1u;
#define S_U32 if (a0+a1+a2+a3+a4+a5+a6+a7 == 12345u) out[0] = 1;
#define B_ADDU32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_add_u32, D_U32, R8(B_ADDU32), S_U32)
#define B_MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_mad_u32_u24, D_U32, R8(B_MADU24), S_U32)
#define B_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_mul_lo_u32, D_U32, R8(B_MULLO), S_U32)
#define B_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_bpermute_wait, D_U32, R8(B_BPERM), S_U32)
#define B_BPERMNW(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a##i) : "v"(c));
#define B_WAITL asm volatile("s_waitcnt lgkmcnt(0)");
DEF_KERNEL(k_bpermute_x8, D_U32, R8(B_BPERMNW) B_WAITL, S_U32)
#define B_DPP(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
DEF_KERNEL(k_add_u32_dpp_row_shr, D_U32, R8(B_DPP), S_U32)
#define B_DPPW(i) asm volatile("v_add_u32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
DEF_KERNEL(k_add_u32_dpp_wave_shr, D_U32, R8(B_DPPW), S_U32)

// ---- u64 / f64
#define D_F64 double a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; double c=seed*0.5;
#define S_F64 if (a0+a1+a2+a3+a4+a5+a6+a7 == 12345.0) out[0] = 1;
#define B_ADDF64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_add_f64, D_F64, R8(B_ADDF64), S_F64)
#define B_MULF64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_mul_f64, D_F64, R8(B_MULF64), S_F64)
#define B_FMAF64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_fma_f64, D_F64, R8(B_FMAF64), S_F64)
#define B_RCPF64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
DEF_KERNEL(k_rcp_f64, D_F64, R8(B_RCPF64), S_F64)
#define B_MINF64(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_min_f64, D_F64, R8(B_MINF64), S_F64)
#define B_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_pk_add_f32, D_F64, R8(B_PKADD), S_F64)
#define B_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(c));
DEF_KERNEL(k_pk_fma_f32, D_F64, R8(B_PKFMA), S_F64)

#define D_MIX double a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; \
              float f0=seed,f1=seed+1,f2=seed+2,f3=seed+3,f4=seed+4,f5=seed+5,f6=seed+6,f7=seed+7; \
              unsigned u0=seed,u1=seed+1,u2=seed+2,u3=seed+3,u4=seed+4,u5=seed+5,u6=seed+6,u7=seed+7;
#define S_MIX if (a0+a1+a2+a3+a4+a5+a6+a7 == 12345.0 || f0+f1+f2+f3+f4+f5+f6+f7 == 3.f || u0+u1+u2+u3+u4+u5+u6+u7 == 77u) out[0] = 1;
#define B_CVTF64F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a##i) : "v"(f##i));
DEF_KERNEL(k_cvt_f64_f32, D_MIX, R8(B_CVTF64F32), S_MIX)
#define B_CVTF64U32(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a##i) : "v"(u##i));
DEF_KERNEL(k_cvt_f64_u32, D_MIX, R8(B_CVTF64U32), S_MIX)
#define B_CVTF32F64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f##i) : "v"(a##i));
DEF_KERNEL(k_cvt_f32_f64, D_MIX, R8(B_CVTF32F64), S_MIX)
#define B_CVTI32F64(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u##i) : "v"(a##i));
DEF_KERNEL(k_cvt_i32_f64, D_MIX, R8(B_CVTI32F64), S_MIX)

#define D_U64 unsigned long long a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; unsigned c=(unsigned)seed*3u+1u; unsigned long long c64 = c * 77ull;
#define S_U64 if (a0+a1+a2+a3+a4+a5+a6+a7 == 12345ull) out[0] = 1;
#define B_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a##i) : "v"(c) : "vcc");
DEF_KERNEL(k_mad_u64_u32, D_U64, R8(B_MAD64), S_U64)
#define B_ADD64(i) a##i += c64; asm volatile("" : "+v"(a##i));
DEF_KERNEL(k_add_u64_pair, D_U64, R8(B_ADD64), S_U64)
#define B_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a##i));
DEF_KERNEL(k_lshl_b64, D_U64, R8(B_LSHL64), S_U64)

// ---- LDS (each wave touches its own 2 KB)
#define D_LDS __shared__ double lds[4 * 256 + 64]; double a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; \
              unsigned addr = threadIdx.x * 8; lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed; __syncthreads();
#define B_LDSR64(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*8" : "=v"(a##i) : "v"(addr));
DEF_KERNEL(k_ds_read_b64_x8, D_LDS, R8(B_LDSR64) B_WAITL, S_F64)
#define B_LDSR2(i) asm volatile("ds_read2_b64 %0, %1 offset0:" #i " offset1:" #i "+9" : "=v"(q##i) : "v"(addr));
#define D_LDS2 __shared__ double lds[4 * 256 + 64]; double2 q0,q1,q2,q3,q4,q5,q6,q7; \
              unsigned addr = threadIdx.x * 8; lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed; __syncthreads();
#define S_LDS2 if (q0.x+q1.x+q2.x+q3.x+q4.y+q5.y+q6.y+q7.y == 12345.0) out[0] = 1;
DEF_KERNEL(k_ds_read2_b64_x8, D_LDS2, R8(B_LDSR2) B_WAITL, S_LDS2)
#define B_LDSW64(i) asm volatile("ds_write_b64 %1, %0 offset:" #i "*8" : : "v"(a##i), "v"(addr));
DEF_KERNEL(k_ds_write_b64_x8, D_LDS, R8(B_LDSW64) B_WAITL, S_F64)
#define B_LDSR128(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16" : "=v"(w##i) : "v"(addr2));
#define D_LDS4 __shared__ double lds[4 * 256 + 64]; float4 w0,w1,w2,w3,w4,w5,w6,w7; \
              unsigned addr2 = threadIdx.x * 16; lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed; __syncthreads();
#define S_LDS4 if (w0.x+w1.x+w2.x+w3.x+w4.y+w5.y+w6.y+w7.y == 12345.f) out[0] = 1;
DEF_KERNEL(k_ds_read_b128_x8, D_LDS4, R8(B_LDSR128) B_WAITL, S_LDS4)
#define B_LDSR32(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*4" : "=v"(u##i) : "v"(addr));
#define D_LDS1 __shared__ double lds[4 * 256 + 64]; unsigned u0,u1,u2,u3,u4,u5,u6,u7; \
              unsigned addr = threadIdx.x * 4; lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed; __syncthreads();
#define S_LDS1 if (u0+u1+u2+u3+u4+u5+u6+u7 == 12345u) out[0] = 1;
DEF_KERNEL(k_ds_read_b32_x8, D_LDS1, R8(B_LDSR32) B_WAITL, S_LDS1)

typedef void (*kern_t)(unsigned long long *, double);
struct Entry { const char *name; kern_t k; };

int main()
{
    Entry es[] = {
        {"v_add_f32", k_add_f32}, {"v_fma_f32", k_fma_f32}, {"v_cvt_u32_f32", k_cvt_u32_f32}, {"v_rcp_f32", k_rcp_f32},
        {"v_add_u32", k_add_u32}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32},
        {"v_add_u32 dpp row_shr", k_add_u32_dpp_row_shr}, {"v_add_u32 dpp wave_shr", k_add_u32_dpp_wave_shr},
        {"ds_bpermute+wait each", k_bpermute_wait}, {"ds_bpermute x8 then wait", k_bpermute_x8},
        {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64}, {"v_min_f64", k_min_f64}, {"v_rcp_f64", k_rcp_f64},
        {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"u64 add (add_co+addc)", k_add_u64_pair}, {"v_lshlrev_b64", k_lshl_b64},
        {"ds_read_b32 x8", k_ds_read_b32_x8}, {"ds_read_b64 x8", k_ds_read_b64_x8}, {"ds_read2_b64 x8", k_ds_read2_b64_x8}, {"ds_read_b128 x8", k_ds_read_b128_x8},
        {"ds_write_b64 x8", k_ds_write_b64_x8},
    };
    unsigned long long *d_out;
    const int maxw = 256 * 4 * 8;
    CHECK(hipMalloc(&d_out, maxw * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(maxw);
    printf("%-28s %10s %10s %10s   (cycles per wave-instruction per SIMD; s_memtime ticks)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (auto &e : es) {
        printf("%-28s", e.name);
        for (int wps : {1, 2, 4}) {
            // blocks of 256 threads = 4 waves = one per SIMD; wps blocks per CU
            const int blocks = 256 * wps;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.5);
            CHECK(hipDeviceSynchronize());
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.5);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), d_out, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < blocks * 4; ++i) sum += (double)h[i];
            const double per_wave = sum / (blocks * 4);
            // per SIMD: wps waves each issued ITERS*8 instructions during ~per_wave cycles
            printf(" %10.2f", per_wave / (ITERS * 8.0 * wps));
        }
        printf("\n");
    }
    return 0;
}
