// Per-instruction integer-VALU throughput on gfx950: which forms run at the SIMD-32 rate?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_microbench tools/valu_microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define OPS "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])

#define KERNEL(NAME, BODY)                                                                      \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t seed) {      \
        uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                   \
        uint32_t r[8];                                                                          \
        for (int k = 0; k < 8; ++k) r[k] = tid * 2654435761u + k * 977u + seed;                \
        uint32_t a = tid ^ seed, b = ~tid * 31u;                                                \
        uint32_t sa = seed * 3u;                                                                \
        for (int i = 0; i < iters; ++i) {                                                       \
            asm volatile(BODY BODY BODY BODY : OPS : "v"(a), "v"(b), "s"(sa));                  \
        }                                                                                       \
        uint32_t s = 0;                                                                         \
        for (int k = 0; k < 8; ++k) s ^= r[k];                                                  \
        out[tid] = s;                                                                           \
    }

#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define I_XOR_S(k) "v_xor_b32 %" #k ", %10, %" #k "\n"
#define I_AND_OR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define I_OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %9\n"
#define I_LSHL_OR(k) "v_lshl_or_b32 %" #k ", %" #k ", 1, %9\n"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_BITOP3(k) "v_bitop3_b32 %" #k ", %" #k ", %8, %9 bitop3:0x80\n"
#define I_BITOP3_S(k) "v_bitop3_b32 %" #k ", %10, %" #k ", %9 bitop3:0x48\n"
#define I_BITOP2(k) "v_bitop3_b32 %" #k ", %" #k ", %8, %8 bitop3:0x80\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define I_ADD(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %8, %9\n"
#define I_PKADD16(k) "v_pk_add_u16 %" #k ", %" #k ", %8\n"
#define I_PKLSHL16(k) "v_pk_lshlrev_b16 %" #k ", 1, %" #k "\n"
#define I_AND_OR_E(k) "v_and_or_b32 %" #k ", %" #k ", %8, 1\n"
#define I_XOR3(k) "v_xad_u32 %" #k ", %" #k ", %8, %9\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_BCNT(k) "v_bcnt_u32_b32 %" #k ", %8, %" #k "\n"
#define I_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 31\n"
#define I_ALIGNBIT_V(k) "v_alignbit_b32 %" #k ", %" #k ", %8, %9\n"
#define I_CMP_ADDC(k) "v_cmp_le_u32 vcc, %" #k ", %8\n" "v_addc_co_u32 %" #k ", vcc, %" #k ", %" #k ", vcc\n"
#define I_CMP(k) "v_cmp_le_u32 vcc, %" #k ", %8\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_LSHR(k) "v_lshrrev_b32 %" #k ", 1, %" #k "\n"
#define I_SUB(k) "v_sub_u32 %" #k ", %" #k ", %8\n"
#define I_OR(k) "v_or_b32 %" #k ", %" #k ", %8\n"
#define I_MIN(k) "v_min_u32 %" #k ", %" #k ", %8\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_ANDN(k) "v_bitop3_b32 %" #k ", %" #k ", %8, %8 bitop3:0x30\n"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 1, %8\n"
#define I_FILTER(k) "v_and_b32 %" #k ", %8, %" #k "\n" "v_bcnt_u32_b32 %" #k ", %" #k ", %9\n" "v_alignbit_b32 %" #k ", %8, %" #k ", 31\n"

#define I_BFEI(k) "v_bfe_i32 %" #k ", %" #k ", 5, 1\n"
#define I_BFEU(k) "v_bfe_u32 %" #k ", %" #k ", 6, 8\n"
#define I_BFEU_V(k) "v_bfe_u32 %" #k ", %8, %" #k ", 1\n"
#define I_ASHR(k) "v_ashrrev_i32 %" #k ", 31, %" #k "\n"
#define I_LSHRV(k) "v_lshrrev_b32 %" #k ", %8, %" #k "\n"
#define I_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
KERNEL(k_bfe_i32, REP8(I_BFEI))
KERNEL(k_bfe_u32, REP8(I_BFEU))
KERNEL(k_bfe_u32_v, REP8(I_BFEU_V))
KERNEL(k_ashr, REP8(I_ASHR))
KERNEL(k_lshr_v, REP8(I_LSHRV))
KERNEL(k_mad24, REP8(I_MAD24))
KERNEL(k_xor, REP8(I_XOR))
KERNEL(k_xor_s, REP8(I_XOR_S))
KERNEL(k_and, REP8(I_AND))
KERNEL(k_add, REP8(I_ADD))
KERNEL(k_lshl, REP8(I_LSHL))
KERNEL(k_and_or, REP8(I_AND_OR))
KERNEL(k_and_or_inl, REP8(I_AND_OR_E))
KERNEL(k_or3, REP8(I_OR3))
KERNEL(k_lshl_or, REP8(I_LSHL_OR))
KERNEL(k_bitop3, REP8(I_BITOP3))
KERNEL(k_bitop3_s, REP8(I_BITOP3_S))
KERNEL(k_bitop3_2src, REP8(I_BITOP2))
KERNEL(k_xad, REP8(I_XOR3))
KERNEL(k_fma, REP8(I_FMA))
KERNEL(k_fmac, REP8(I_FMAC))
KERNEL(k_pk_add_u16, REP8(I_PKADD16))
KERNEL(k_pk_lshl_b16, REP8(I_PKLSHL16))
KERNEL(k_mov, REP8(I_MOV))
KERNEL(k_bcnt, REP8(I_BCNT))
KERNEL(k_alignbit, REP8(I_ALIGNBIT))
KERNEL(k_alignbit_v, REP8(I_ALIGNBIT_V))
KERNEL(k_cmp_addc, REP8(I_CMP_ADDC))
KERNEL(k_cmp, REP8(I_CMP))
KERNEL(k_cndmask, REP8(I_CNDMASK))
KERNEL(k_lshr, REP8(I_LSHR))
KERNEL(k_sub, REP8(I_SUB))
KERNEL(k_or, REP8(I_OR))
KERNEL(k_min, REP8(I_MIN))
KERNEL(k_perm, REP8(I_PERM))
KERNEL(k_andn, REP8(I_ANDN))
KERNEL(k_add3, REP8(I_ADD3))
KERNEL(k_lshladd, REP8(I_LSHLADD))
KERNEL(k_filter, REP8(I_FILTER))
// alternating VOP2 / VOP3
#define I_MIX(k) "v_xor_b32 %" #k ", %" #k ", %8\n" "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
KERNEL(k_mix_xor_andor, REP8(I_MIX))

typedef void (*kern_t)(uint32_t *, int, uint32_t);
struct Case { const char *name; kern_t fn; int per_body; };

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    std::vector<Case> cases = {
        {"v_xor_b32 (VOP2)", k_xor, 8}, {"v_xor_b32 sgpr src0", k_xor_s, 8}, {"v_and_b32", k_and, 8}, {"v_add_u32", k_add, 8},
        {"v_lshlrev_b32", k_lshl, 8}, {"v_and_or_b32", k_and_or, 8}, {"v_and_or_b32 inline const", k_and_or_inl, 8},
        {"v_or3_b32", k_or3, 8}, {"v_lshl_or_b32", k_lshl_or, 8}, {"v_bitop3_b32 3 vgpr", k_bitop3, 8},
        {"v_bitop3_b32 sgpr+2vgpr", k_bitop3_s, 8}, {"v_bitop3_b32 2 distinct vgpr", k_bitop3_2src, 8}, {"v_xad_u32", k_xad, 8},
        {"v_fma_f32", k_fma, 8}, {"v_fmac_f32 (VOP2)", k_fmac, 8}, {"v_pk_add_u16", k_pk_add_u16, 8},
        {"v_pk_lshlrev_b16", k_pk_lshl_b16, 8}, {"v_mov_b32", k_mov, 8}, {"mix xor+and_or", k_mix_xor_andor, 16},
        {"v_bcnt_u32_b32", k_bcnt, 8}, {"v_alignbit_b32 imm shift", k_alignbit, 8}, {"v_alignbit_b32 vgpr shift", k_alignbit_v, 8},
        {"v_cmp_le_u32 + v_addc_co_u32", k_cmp_addc, 16}, {"v_cmp_le_u32 vcc", k_cmp, 8}, {"v_cndmask_b32 vcc", k_cndmask, 8},
        {"v_lshrrev_b32", k_lshr, 8}, {"v_sub_u32", k_sub, 8}, {"v_or_b32", k_or, 8}, {"v_min_u32", k_min, 8},
        {"v_perm_b32", k_perm, 8}, {"v_bitop3 andn 2src", k_andn, 8}, {"v_add3_u32", k_add3, 8}, {"v_lshl_add_u32", k_lshladd, 8},
        {"filter mix and+bcnt+alignbit", k_filter, 24},
        {"v_bfe_i32 imm", k_bfe_i32, 8}, {"v_bfe_u32 imm", k_bfe_u32, 8}, {"v_bfe_u32 vgpr offset", k_bfe_u32_v, 8},
        {"v_ashrrev_i32 imm", k_ashr, 8}, {"v_lshrrev_b32 vgpr shift", k_lshr_v, 8}, {"v_mad_u32_u24", k_mad24, 8}};
    const int threads = 256, iters = 4000;
    uint32_t *d;
    hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc : {8, 2}) {           // blocks per CU: 8 -> 8 waves/SIMD ... 1 -> 1 wave/SIMD
        printf("---- %d waves per SIMD ----\n", bpc);
        int blocks = p.multiProcessorCount * bpc;
        for (auto &c : cases) {
            c.fn<<<blocks, threads>>>(d, iters / 4, 1);
            hipEventRecord(e0);
            c.fn<<<blocks, threads>>>(d, iters, 2);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double ops = (double)blocks * threads * iters * c.per_body * 4;
            double rate = ops / (ms * 1e-3);
            // cycles per wave-instruction per SIMD at 2.4 GHz
            double wave_instr = ops / 64.0;
            double cyc = (ms * 1e-3) * 2.4e9 * (p.multiProcessorCount * 4) / wave_instr;
            printf("%-32s %7.2f T lane-ops/s  %5.2f cyc/wave-instr/SIMD @2.4GHz  (%.3f ms)\n", c.name, rate / 1e12, cyc, ms);
        }
    }
    return 0;
}
