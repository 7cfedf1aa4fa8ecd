// What scalar instructions and branches cost BESIDE vector instructions on gfx950 (MI355X), wave64, 8 waves per SIMD: the
// loop of k_descriptor issues 29 scalar instructions (exec-mask bookkeeping of its conditional atomics, loop control) and ~7
// branches per 97 vector instructions.  Each kernel repeats a group of 8 v_fma_f32 plus S scalar instructions / B taken
// branches; the table gives SIMD cycles per GROUP (ticks of one wave / groups of all waves on its SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/salu_mix.hip -o tools/ubench/salu_mix && tools/ubench/salu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 2048;
#define V8 "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n" \
           "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
#define S1 "s_add_u32 s20, s20, 1\n"
#define S2 S1 "s_and_b32 s21, s21, s20\n"
#define S4 S2 "s_xor_b32 s22, s22, s21\n s_or_b32 s23, s23, s20\n"
#define S8 S4 "s_add_u32 s24, s24, 3\n s_and_b32 s25, s25, s24\n s_xor_b32 s26, s26, s25\n s_or_b32 s27, s27, s24\n"
// a taken branch to the next instruction (the scalar compare makes it depend on a register the assembler cannot fold)
#define B1(L) "s_cmp_lg_u32 s20, 0x7fffffff\n s_cbranch_scc1 " L "\n s_nop 0\n" L ":\n"
// a branch that is NOT taken (exec is never empty here): what the compiler puts around every conditional block
#define N1(L) "s_cbranch_execz " L "\n"
// ... and the whole idiom of a conditional block: mask, skip if empty, (block), restore
#define C1(L) "s_and_saveexec_b64 s[28:29], exec\n s_cbranch_execz " L "\n" L ":\n s_or_b64 exec, exec, s[28:29]\n"
#define E1 "s_and_saveexec_b64 s[28:29], vcc\n s_or_b64 exec, exec, s[28:29]\n"

#define KERNEL(NAME, INS)                                                                                                  \
    __global__ __launch_bounds__(64) void NAME(long long* out, float a, float b)                                           \
    {                                                                                                                      \
        float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;  \
        const long long t0 = __builtin_readcyclecounter();                                                                \
        for (int it = 0; it < ITERS; it++)                                                                                 \
            asm volatile(INS : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                 \
                         : "v"(a), "v"(b)                                                                                  \
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc", "vcc");              \
        const long long t1 = __builtin_readcyclecounter();                                                                \
        if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = 1;                                               \
        if (threadIdx.x == 0) out[1 + blockIdx.x] = t1 - t0;                                                               \
    }
KERNEL(k_v8, V8)
KERNEL(k_v8s1, V8 S1)
KERNEL(k_v8s2, V8 S2)
KERNEL(k_v8s4, V8 S4)
KERNEL(k_v8s8, V8 S8)
KERNEL(k_s8, S8)
KERNEL(k_v8b1, V8 B1("Lsm_a%="))
KERNEL(k_v8b2, V8 B1("Lsm_a%=") B1("Lsm_b%="))
KERNEL(k_v8e1, V8 E1)
KERNEL(k_v8e2, V8 E1 E1)
KERNEL(k_v8s2b1, V8 S2 B1("Lsm_a%="))
KERNEL(k_v8n1, V8 N1("Lsm_z%=") "Lsm_z%=:\n")
KERNEL(k_v8n2, V8 N1("Lsm_z%=") N1("Lsm_z%=") "Lsm_z%=:\n")
KERNEL(k_v8n4, V8 N1("Lsm_z%=") N1("Lsm_z%=") N1("Lsm_z%=") N1("Lsm_z%=") "Lsm_z%=:\n")
KERNEL(k_v8c1, V8 C1("Lsm_c%="))
KERNEL(k_v8c2, V8 C1("Lsm_c%=") C1("Lsm_d%="))
KERNEL(k_v8c4, V8 C1("Lsm_c%=") C1("Lsm_d%=") C1("Lsm_e%=") C1("Lsm_f%="))

typedef void (*kern_t)(long long*, float, float);
int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    long long* d;
    hipMalloc(&d, sizeof(long long) * (1 + cus * 64));
    struct { const char* name; kern_t k; } ks[] = {
        {"8 v_fma", k_v8}, {"8 v_fma + 1 salu", k_v8s1}, {"8 v_fma + 2 salu", k_v8s2}, {"8 v_fma + 4 salu", k_v8s4},
        {"8 v_fma + 8 salu", k_v8s8}, {"8 salu alone", k_s8}, {"8 v_fma + 1 cmp/taken branch", k_v8b1},
        {"8 v_fma + 2 cmp/taken branch", k_v8b2}, {"8 v_fma + 1 saveexec/restore", k_v8e1}, {"8 v_fma + 2 saveexec/restore", k_v8e2},
        {"8 v_fma + 2 salu + 1 branch", k_v8s2b1},
        {"8 v_fma + 1 branch not taken", k_v8n1}, {"8 v_fma + 2 branches not taken", k_v8n2}, {"8 v_fma + 4 branches not taken", k_v8n4},
        {"8 v_fma + 1 mask/skip/restore", k_v8c1}, {"8 v_fma + 2 mask/skip/restore", k_v8c2}, {"8 v_fma + 4 mask/skip/restore", k_v8c4}};
    printf("%-34s %s\n", "group", "SIMD cycles per group at 1 / 2 / 4 / 8 waves per SIMD (all CUs busy)");
    for (auto& e : ks) {
        printf("%-34s", e.name);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * 4 * wps;
            hipMemset(d, 0, sizeof(long long) * (1 + blocks));
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.9999f);
            hipDeviceSynchronize();
            std::vector<long long> h(1 + blocks);
            hipMemcpy(h.data(), d, sizeof(long long) * (1 + blocks), hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < blocks; i++) sum += (double)h[1 + i];
            printf(" %7.2f", sum / blocks / (double)((long long)ITERS * wps));
        }
        printf("\n");
    }
    hipFree(d);
    return 0;
}
