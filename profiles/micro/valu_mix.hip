// VALU issue-cost probe by instruction kind: W waves per SIMD (W workgroups of 256 threads per CU), each wave
// running a long stream of independent instructions of ONE kind (inline asm, eight independent register chains).
// Prints cycles per wave-instruction per SIMD (= SIMD cycles / instructions issued on that SIMD by all its waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define K_FMA(i)    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_ADD(i)    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_MUL(i)    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_MIN(i)    asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_MAX3(i)   asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_MOV(i)    asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(a));
#define K_AND(i)    asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_LSHL(i)   asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]));
#define K_ADDU(i)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_BFE(i)    asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x[i]));
#define K_CMP(i)    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
#define K_CMPS(i)   asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(x[i]), "v"(a) : "s20", "s21");
#define K_CND(i)    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
#define K_CND64(i)  asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(x[i]) : "v"(a));
#define K_CNDK(i)   asm volatile("v_cndmask_b32_e64 %0, 0, 4, s[22:23]" : "=v"(x[i]));
#define K_MAXU(i)   asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_OR3(i)    asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(a));
#define K_OR(i)     asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_LSHLV(i)  asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
#define K_CMPU(i)   asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
#define K_FMAC(i)   asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_MULK(i)   asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x[i]));
#define K_ADDS(i)   asm volatile("v_add_f32 %0, s24, %0" : "+v"(x[i]));
#define K_SUB(i)    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_MAXF(i)   asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_ASHR(i)   asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x[i]));
#define K_BPERM(i)  asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x[i]) : "v"(a));
#define K_MAD64(i)  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(z[i]) : "v"(a), "v"(b) : "vcc");
#define K_MULHI(i)  asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_DIVSC(i)  asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x[i]) : "v"(a) : "vcc");
#define K_DIVFM(i)  asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_DIVFX(i)  asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_SQRT(i)   asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
#define K_RSQ(i)    asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
#define K_EXP(i)    asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#define K_LOG(i)    asm volatile("v_log_f32 %0, %0" : "+v"(x[i]));
#define K_SIN(i)    asm volatile("v_sin_f32 %0, %0" : "+v"(x[i]));
#define K_FRACT(i)  asm volatile("v_fract_f32 %0, %0" : "+v"(x[i]));
#define K_LDEXP(i)  asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_RDLANE(i) asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(x[i]) : "s20");
#define K_WRLANE(i) asm volatile("v_writelane_b32 %0, s24, 5" : "+v"(x[i]));
#define K_CVTUB(i)  asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x[i]));
#define K_CVTI(i)   asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x[i]));
#define K_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(z[i]));
#define K_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(z[i]) : "v"(z[(i+1)&7]));
#define K_MOV64(i)  asm volatile("v_mov_b64 %0, %1" : "=v"(z[i]) : "v"(z[(i+1)&7]));
#define K_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x36" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_DPP(i)    asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
#define K_CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
#define K_PKMUL(i)  asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(c));
#define K_PKADD(i)  asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(c));
#define K_PKFMA(i)  asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(c));
#define K_RCP(i)    asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
#define K_MULLO(i)  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_MAD24(i)  asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_BFI(i)    asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_MED3(i)   asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_SUBREV(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
#define K_XOR(i)    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_ADD3(i)   asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define K_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(a));
#define K_BCNT(i)   asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define K_FFBL(i)   asm volatile("v_ffbl_b32 %0, %0" : "+v"(x[i]));
#define K_CVT(i)    asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]));
#define K_MBCNT(i)  asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
#define K_READFL(i) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(x[i]) : "s20");
#define K_SALU(i)   asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");

typedef float f2 __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, BODY)                                                                              \
    __global__ void __launch_bounds__(256) k_##NAME(float *out, int iters, float a_, float b_) {         \
        float x[8];                                                                                     \
        f2 y[8];                                                                                        \
        unsigned long long z[8];                                                                        \
        for (int i = 0; i < 8; i++) { x[i] = threadIdx.x + i; y[i] = f2{x[i], x[i] + 1.f}; z[i] = threadIdx.x * 77ull + i; }            \
        float a = a_, b = b_;                                                                           \
        f2 c = {a_, b_};                                                                                \
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));                                                   \
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_mov_b64 s[22:23], vcc\n s_mov_b32 s24, 0x3f000000" : : "v"(a), "v"(b) : "vcc", "s22", "s23", "s24");                                                  \
        for (int it = 0; it < iters; it++) {                                                            \
            REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY)     \
        }                                                                                               \
        float s = 0;                                                                                    \
        for (int i = 0; i < 8; i++) s += x[i] + y[i].x + y[i].y + (float)z[i];                                        \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                        \
    }

#define LIST(F) F(FMA, 1) F(ADD, 1) F(MUL, 1) F(SUBREV, 1) F(MIN, 1) F(MAX3, 1) F(MED3, 1) F(MOV, 1) F(AND, 1) F(XOR, 1) F(LSHL, 1) F(ADDU, 1) \
    F(ADD3, 1) F(LSHLADD, 1) F(BFE, 1) F(BFI, 1) F(BCNT, 1) F(FFBL, 1) F(CVT, 1) F(MBCNT, 1) F(CMP, 1) F(CMPS, 1) F(CMPU, 1) F(CND, 1) F(CND64, 1) F(CNDK, 1) F(CMPCND, 2) F(MAXU, 1) F(MAXF, 1) F(OR, 1) F(OR3, 1) F(LSHLOR, 1) F(LSHLV, 1) \
    F(FMAC, 1) F(MULK, 1) F(ADDS, 1) F(SUB, 1) F(ASHR, 1) F(BPERM, 1) F(DPP, 1) \
    F(MAD64, 1) F(MULHI, 1) F(DIVSC, 1) F(DIVFM, 1) F(DIVFX, 1) F(SQRT, 1) F(RSQ, 1) F(EXP, 1) F(LOG, 1) F(SIN, 1) F(FRACT, 1) F(LDEXP, 1) F(RDLANE, 1) \
    F(WRLANE, 1) F(CVTUB, 1) F(CVTI, 1) F(LSHL64, 1) F(LSHLADD64, 1) F(MOV64, 1) F(BITOP3, 1)      \
    F(PKMUL, 1) F(PKADD, 1) F(PKFMA, 1) F(RCP, 1) F(MULLO, 1) F(MAD24, 1) F(READFL, 1) F(SALU, 1)

#define DEF(NAME, N) KERNEL(NAME, K_##NAME)
LIST(DEF)

struct Entry { const char *name; void (*fn)(float *, int, float, float); int per; };
#define ENT(NAME, N) {#NAME, k_##NAME, N},
static Entry entries[] = {LIST(ENT)};

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * cus * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const double clk = p.clockRate * 1e3;
    printf("clock %.0f MHz, %d CUs; cycles per wave-instruction per SIMD at W waves per SIMD\n", clk / 1e6, cus);
    printf("%-8s", "kind");
    for (int w = 1; w <= 6; w++) printf("  W=%d  ", w);
    printf("\n");
    for (const Entry &e : entries) {
        if (argc > 1 && strcmp(argv[1], e.name)) continue;
        printf("%-8s", e.name);
        for (int w = 1; w <= 6; w++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(e.fn, dim3(cus * w), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double insts = (double)iters * 64 * e.per * w;      // wave-instructions issued per SIMD
            printf("  %5.2f ", ms * 1e-3 * clk / insts);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
