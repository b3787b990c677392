// Does a VALU write to the data registers of a 16-byte buffer store, issued right behind the store, reach memory on gfx950?
// Round 2 saw k_guided_pipe's one 16-byte store per lane (register soffset) pick up the next instruction's write to its data
// registers and has used two 8-byte stores since.  The ISA manuals of this family list the case --
// "VMEM store of more than 64 bits followed by a VALU write of the store's data VGPRs: 1 wait state" -- and LLVM's hazard
// recognizer applies it only when soffset is NOT a register.  This program issues the pair from inline assembly (nothing is
// inserted inside one asm statement) for {16-byte, 8-byte} x {soffset in an SGPR, soffset = 0} x {0, 1, 2 wait states} and
// counts the dwords in memory that hold the poison the following v_mov wrote.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/microbench/store_hazard profiles/microbench/store_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef int __attribute__((ext_vector_type(4))) i32x4;
constexpr uint32_t kPoison = 0xbad00000u;
constexpr int kIters = 64;

#define NOP0 ""
#define NOP1 "s_nop 0\n"
#define NOP2 "s_nop 1\n"

#define DEF(NAME, STORE, SOFF, NOPS)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t bytes)                                         \
    {                                                                                                                   \
        const uint64_t a = (uint64_t)out;                                                                               \
        i32x4 r;                                                                                                        \
        r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);                                                         \
        r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32) & 0xffff);                                        \
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);                                                               \
        r.w = 0x00020000;                                                                                               \
        const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;                         \
        for (int it = 0; it < kIters; ++it) {                                                                           \
            const uint32_t slot = (wave * kIters + (uint32_t)it) * 64u + lane;  /* 16 bytes per slot */                 \
            const uint32_t off = lane * 16u;                                                                            \
            const int soff = __builtin_amdgcn_readfirstlane((int)((wave * kIters + (uint32_t)it) * 1024u));              \
            const uint32_t tag = slot * 4u;                                                                             \
            asm volatile("v_mov_b32 v10, %[d0]\n v_add_u32 v11, 1, %[d0]\n v_add_u32 v12, 2, %[d0]\n v_add_u32 v13, 3, %[d0]\n"   \
                         "s_nop 4\n" STORE NOPS                                                                         \
                         "v_mov_b32 v10, %[p]\n v_mov_b32 v11, %[p]\n v_mov_b32 v12, %[p]\n v_mov_b32 v13, %[p]\n"      \
                         ::[d0] "v"(tag), [off] "v"(SOFF == 1 ? off : off + (uint32_t)soff), [rsrc] "s"(r), [soff] "s"(soff),          \
                         [p] "v"(kPoison)                                                                               \
                         : "v10", "v11", "v12", "v13", "memory");                                                      \
        }                                                                                                               \
    }

#define ST16_REG "buffer_store_dwordx4 v[10:13], %[off], %[rsrc], %[soff] offen\n"
#define ST16_IMM "buffer_store_dwordx4 v[10:13], %[off], %[rsrc], 0 offen\n"
#define ST8_REG "buffer_store_dwordx2 v[10:11], %[off], %[rsrc], %[soff] offen\n buffer_store_dwordx2 v[12:13], %[off], %[rsrc], %[soff] offen offset:8\n"

DEF(k16_reg_0, ST16_REG, 1, NOP0)
DEF(k16_reg_1, ST16_REG, 1, NOP1)
DEF(k16_reg_2, ST16_REG, 1, NOP2)
DEF(k16_imm_0, ST16_IMM, 0, NOP0)
DEF(k16_imm_1, ST16_IMM, 0, NOP1)
DEF(k8_reg_0, ST8_REG, 1, NOP0)

int main()
{
    const int nblk = 2048, nwave = nblk * 4;
    const size_t ndw = (size_t)nwave * kIters * 64 * 4;
    uint32_t *d;
    CHECK(hipMalloc(&d, ndw * 4));
    std::vector<uint32_t> h(ndw);
    struct V { const char *name; void (*k)(uint32_t *, uint32_t); } vs[] = {
        {"16-byte store, soffset in an SGPR, v_mov right behind ", k16_reg_0}, {"16-byte store, soffset in an SGPR, s_nop 0 between    ", k16_reg_1},
        {"16-byte store, soffset in an SGPR, s_nop 1 between    ", k16_reg_2}, {"16-byte store, soffset = 0,        v_mov right behind ", k16_imm_0},
        {"16-byte store, soffset = 0,        s_nop 0 between    ", k16_imm_1}, {"two 8-byte stores, soffset in an SGPR, v_mov behind   ", k8_reg_0}};
    for (const V &v : vs) {
        CHECK(hipMemset(d, 0, ndw * 4));
        hipLaunchKernelGGL(v.k, dim3(nblk), dim3(256), 0, 0, d, (uint32_t)(ndw * 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d, ndw * 4, hipMemcpyDeviceToHost));
        size_t poison = 0, wrong = 0, which[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < ndw; ++i) {
            if (h[i] == kPoison) { ++poison; ++which[i & 3]; }
            else if (h[i] != (uint32_t)i) ++wrong;
        }
        printf("%s: %zu of %zu dwords hold the poison (dword 0..3 of the store: %zu %zu %zu %zu), %zu other mismatches\n", v.name, poison, ndw,
               which[0], which[1], which[2], which[3], wrong);
    }
    return 0;
}
