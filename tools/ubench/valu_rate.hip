// Issue rate of single vector instructions on gfx950 (MI355X), wave64: cycles a SIMD needs per instruction when W waves
// issue independent copies of it back to back.  The microarchitecture guide gives 2 cycles for v_fma_f32; the descriptor
// kernel is made of compares, selects, conversions and integer address arithmetic as much as of FMAs, and its budget
// depends on what THOSE cost.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate && tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define BODY(INS)                                                                                           \
    for (int it = 0; it < ITERS; it++) {                                                                    \
        REP8(asm volatile(INS : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                              : "v"(a), "v"(b)                                                              \
                              : "vcc");)                                                                    \
    }

constexpr int ITERS = 256; /* x 8 statements x 8 instructions = 16384 instructions per wave */

#define KERNEL(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                                     \
    __global__ __launch_bounds__(64) void NAME(long long* out, float a, float b)                                          \
    {                                                                                                                     \
        float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        const long long t0 = __builtin_readcyclecounter();                                                               \
        BODY(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7)                                                  \
        const long long t1 = __builtin_readcyclecounter();                                                               \
        if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = 1;                                              \
        if (threadIdx.x == 0) out[1 + blockIdx.x] = t1 - t0;                                                              \
    }
#define K1(NAME, FMT) KERNEL(NAME, FMT(0), FMT(1), FMT(2), FMT(3), FMT(4), FMT(5), FMT(6), FMT(7))

#define F_FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i
#define F_FMAC(i) "v_fmac_f32 %" #i ", %8, %9"
#define F_MUL(i) "v_mul_f32 %" #i ", %8, %" #i
#define F_ADD(i) "v_add_f32 %" #i ", %8, %" #i
#define F_MAX3(i) "v_max3_f32 %" #i ", %8, %9, %" #i
#define F_MIN(i) "v_min_f32 %" #i ", %8, %" #i
#define F_FLOOR(i) "v_floor_f32 %" #i ", %" #i
#define F_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i
#define F_CVTF(i) "v_cvt_f32_i32 %" #i ", %" #i
#define F_CVTU(i) "v_cvt_u32_f32 %" #i ", %" #i
#define F_CNDM(i) "v_cndmask_b32 %" #i ", %8, %" #i ", vcc"
#define F_CMP(i) "v_cmp_lt_f32 vcc, %8, %" #i
#define F_CMPS(i) "v_cmp_lt_f32 s[20:21], %8, %" #i
#define F_ADDU(i) "v_add_u32 %" #i ", %8, %" #i
#define F_ADD3(i) "v_add3_u32 %" #i ", %8, %9, %" #i
#define F_LSHLADD(i) "v_lshl_add_u32 %" #i ", %8, 2, %" #i
#define F_ADDLSHL(i) "v_add_lshl_u32 %" #i ", %8, %" #i ", 3"
#define F_MUL24(i) "v_mul_i32_i24 %" #i ", %8, %" #i
#define F_MAD24(i) "v_mad_u32_u24 %" #i ", %8, %9, %" #i
#define F_MULLO(i) "v_mul_lo_u32 %" #i ", %8, %" #i
#define F_AND(i) "v_and_b32 %" #i ", %8, %" #i
#define F_XOR(i) "v_xor_b32 %" #i ", %8, %" #i
#define F_BFI(i) "v_bfi_b32 %" #i ", %8, %9, %" #i
#define F_ASHR(i) "v_ashrrev_i32 %" #i ", 16, %" #i
#define F_LSHL(i) "v_lshlrev_b32 %" #i ", 5, %" #i
#define F_MOV(i) "v_mov_b32 %" #i ", %8"
#define F_SQRT(i) "v_sqrt_f32 %" #i ", %" #i
#define F_RCP(i) "v_rcp_f32 %" #i ", %" #i
#define F_EXP(i) "v_exp_f32 %" #i ", %" #i
#define F_FMAAK(i) "v_fmaak_f32 %" #i ", %8, %" #i ", 0x3e7c5661"
#define F_SUBABS(i) "v_sub_f32_e64 %" #i ", 1.0, |%" #i "|"
#define F_CMPABS(i) "v_cmp_lt_f32_e64 s[20:21], |%" #i "|, %8"
#define F_OR(i) "v_or_b32 %" #i ", %8, %" #i
#define F_LSHR(i) "v_lshrrev_b32 %" #i ", 5, %" #i
#define F_SUBU(i) "v_sub_u32 %" #i ", %8, %" #i
#define F_SUBF(i) "v_sub_f32 %" #i ", %8, %" #i
#define F_MINI(i) "v_min_i32 %" #i ", %8, %" #i
#define F_FRACT(i) "v_fract_f32 %" #i ", %" #i
#define F_TRUNC(i) "v_trunc_f32 %" #i ", %" #i
#define F_CVTRPI(i) "v_cvt_rpi_i32_f32 %" #i ", %" #i
#define F_CVTFLR(i) "v_cvt_flr_i32_f32 %" #i ", %" #i
#define F_CVTUB(i) "v_cvt_f32_ubyte0 %" #i ", %" #i
#define F_CNDS(i) "v_cndmask_b32 %" #i ", %8, %" #i ", s[22:23]"
#define F_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5"
#define F_MULU24(i) "v_mul_u32_u24 %" #i ", %8, %" #i
#define F_NOT(i) "v_not_b32 %" #i ", %" #i
#define F_MED3(i) "v_med3_f32 %" #i ", %8, %9, %" #i
#define F_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", 2"
#define F_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"

K1(k_fma, F_FMA) K1(k_fmac, F_FMAC) K1(k_mul, F_MUL) K1(k_add, F_ADD) K1(k_max3, F_MAX3) K1(k_min, F_MIN) K1(k_floor, F_FLOOR)
K1(k_cvti, F_CVTI) K1(k_cvtf, F_CVTF) K1(k_cvtu, F_CVTU) K1(k_cndmask, F_CNDM) K1(k_cmp_vcc, F_CMP) K1(k_cmp_sgpr, F_CMPS)
K1(k_addu, F_ADDU) K1(k_add3, F_ADD3) K1(k_lshl_add, F_LSHLADD) K1(k_add_lshl, F_ADDLSHL) K1(k_mul24, F_MUL24)
K1(k_mad24, F_MAD24) K1(k_mullo, F_MULLO) K1(k_and, F_AND) K1(k_xor, F_XOR) K1(k_bfi, F_BFI) K1(k_ashr, F_ASHR)
K1(k_lshl, F_LSHL) K1(k_mov, F_MOV) K1(k_sqrt, F_SQRT) K1(k_rcp, F_RCP) K1(k_exp, F_EXP) K1(k_fmaak, F_FMAAK)
K1(k_sub_abs, F_SUBABS) K1(k_cmp_abs, F_CMPABS)
K1(k_or, F_OR) K1(k_lshr, F_LSHR) K1(k_subu, F_SUBU) K1(k_subf, F_SUBF) K1(k_mini, F_MINI) K1(k_fract, F_FRACT) K1(k_trunc, F_TRUNC)
K1(k_cvtrpi, F_CVTRPI) K1(k_cvtflr, F_CVTFLR) K1(k_cvtub, F_CVTUB) K1(k_cnds, F_CNDS) K1(k_bfe, F_BFE) K1(k_mulu24, F_MULU24)
K1(k_not, F_NOT) K1(k_med3, F_MED3) K1(k_ldexp, F_LDEXP) K1(k_movdpp, F_MOVDPP)
// packed f32: two lanes' worth of work per instruction on register pairs
typedef float v2f __attribute__((ext_vector_type(2)));
#define PKBODY(INS)                                                                                        \
    for (int it = 0; it < ITERS; it++) {                                                                   \
        REP8(asm volatile(INS "\n" INS "\n" INS "\n" INS "\n" INS "\n" INS "\n" INS "\n" INS : "+v"(q0) : "v"(qa)); ) \
    }
#define PKKERNEL(NAME, INS)                                                                      \
    __global__ __launch_bounds__(64) void NAME(long long* out, float a, float b)                  \
    {                                                                                             \
        v2f q0 = {(float)threadIdx.x, a}, qa = {a, b};                                            \
        const long long t0 = __builtin_readcyclecounter();                                       \
        PKBODY(INS)                                                                               \
        const long long t1 = __builtin_readcyclecounter();                                       \
        if (q0.x + q0.y == 12345.678f) out[0] = 1;                                                \
        if (threadIdx.x == 0) out[1 + blockIdx.x] = t1 - t0;                                      \
    }
PKKERNEL(k_pkfma, "v_pk_fma_f32 %0, %1, %1, %0") PKKERNEL(k_pkmul, "v_pk_mul_f32 %0, %1, %0") PKKERNEL(k_pkadd, "v_pk_add_f32 %0, %1, %0")
// a typical mix: compare -> select (vcc dependency inside the wave)
KERNEL(k_cmp_cnd, F_CMP(0), F_CNDM(0), F_CMP(1), F_CNDM(1), F_CMP(2), F_CNDM(2), F_CMP(3), F_CNDM(3))
KERNEL(k_fma_int, F_FMA(0), F_ADDU(1), F_FMA(2), F_ADDU(3), F_FMA(4), F_ADDU(5), F_FMA(6), F_ADDU(7))

typedef void (*kern_t)(long long*, float, float);

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    long long* d;
    hipMalloc(&d, sizeof(long long) * (1 + cus * 64));
    struct { const char* name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_max3_f32", k_max3}, {"v_min_f32", k_min},
        {"v_floor_f32", k_floor}, {"v_cvt_i32_f32", k_cvti}, {"v_cvt_f32_i32", k_cvtf}, {"v_cvt_u32_f32", k_cvtu},
        {"v_cndmask_b32 (vcc)", k_cndmask}, {"v_cmp_lt_f32 -> vcc", k_cmp_vcc}, {"v_cmp_lt_f32 -> sgpr pair", k_cmp_sgpr},
        {"v_cmp_lt_f32 |abs| -> sgpr", k_cmp_abs}, {"v_add_u32", k_addu}, {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshl_add},
        {"v_add_lshl_u32", k_add_lshl}, {"v_mul_i32_i24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo},
        {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_bfi_b32", k_bfi}, {"v_ashrrev_i32", k_ashr}, {"v_lshlrev_b32", k_lshl},
        {"v_mov_b32", k_mov}, {"v_sqrt_f32", k_sqrt}, {"v_rcp_f32", k_rcp}, {"v_exp_f32", k_exp}, {"v_fmaak_f32 (literal)", k_fmaak},
        {"v_sub_f32 1.0 - |x| (VOP3)", k_sub_abs}, {"pair: v_cmp -> v_cndmask", k_cmp_cnd}, {"pair: v_fma_f32 + v_add_u32", k_fma_int},
        {"v_or_b32", k_or}, {"v_lshrrev_b32", k_lshr}, {"v_sub_u32", k_subu}, {"v_sub_f32", k_subf}, {"v_min_i32", k_mini},
        {"v_fract_f32", k_fract}, {"v_trunc_f32", k_trunc}, {"v_cvt_rpi_i32_f32", k_cvtrpi}, {"v_cvt_flr_i32_f32", k_cvtflr},
        {"v_cvt_f32_ubyte0", k_cvtub}, {"v_cndmask_b32 (sgpr mask)", k_cnds}, {"v_bfe_u32", k_bfe}, {"v_mul_u32_u24", k_mulu24},
        {"v_not_b32", k_not}, {"v_med3_f32", k_med3}, {"v_ldexp_f32", k_ldexp}, {"v_mov_b32 dpp quad_perm", k_movdpp},
        {"v_pk_fma_f32 (dependent chain)", k_pkfma}, {"v_pk_mul_f32 (dependent chain)", k_pkmul}, {"v_pk_add_f32 (dependent chain)", k_pkadd}};
    const long long n_ins = (long long)ITERS * 64;
    printf("%-30s %s\n", "instruction", "SIMD cycles per wave64 instruction at 1 / 2 / 4 / 8 waves per SIMD");
    for (auto& e : ks) {
        printf("%-30s", e.name);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * 4 * wps; /* single-wave workgroups: wps waves on every SIMD */
            hipMemset(d, 0, sizeof(long long) * (1 + blocks));
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.9999f);
            hipDeviceSynchronize();
            std::vector<long long> h(1 + blocks);
            hipMemcpy(h.data(), d, sizeof(long long) * (1 + blocks), hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < blocks; i++) sum += (double)h[1 + i];
            /* a wave's span holds the instructions of the wps waves sharing its SIMD */
            printf(" %7.2f", sum / blocks / (double)(n_ins * wps));
        }
        printf("\n");
    }
    hipFree(d);
    return 0;
}
