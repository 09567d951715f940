// issue_cost.hip -- VALU issue cost per instruction class on gfx950, in THROUGHPUT mode:
// every wave runs 8 independent chains of one instruction (or a fixed mix), W waves per
// SIMD (workgroup = 4 W waves, one workgroup per CU, `grid` CUs busy), and the figure
// printed is shader cycles per wave-instruction per SIMD = ticks / (instructions per wave x W).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/issue_cost tools/issue_cost.hip && tools/issue_cost
//
// Used to price the sample loop of k_lnlike: which classes cost
// 4 cycles (the fp64 pipe, 16 lanes per clock), which 2 (32-bit, once two or more waves
// share a SIMD), which more (quarter-rate integer multiplies, transcendentals), and
// whether an fp64 and a 32-bit instruction of different waves overlap.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind {
    K_FMA64, K_MUL64, K_ADD64, K_MIN64, K_LDEXP64, K_RCP64, K_CMP64, K_FMA64_SGPR, K_FMA64_LIT,
    K_FMA32, K_PKFMA32, K_ADD_U32, K_AND_B32, K_LSHL_ADD_U32, K_ASHR_I32, K_CNDMASK, K_MOV_B32,
    K_MOV_B64, K_MOV_DPP, K_EXP32, K_RCP32, K_CVT_F64_I32, K_MUL_LO_U32, K_MAD_U64_U32,
    K_LSHL_ADD_U64, K_DS_READ_B64, K_DS_READ_B128, K_MIX_FMA64_ADDU32, K_MIX_FMA64_2ADDU32,
    K_MIX_FMA64_CNDMASK, K_MIX_FMA64_LDEXP, K_MIX_FMA64_DS128, K_SAMPLE_MIX, K_COUNT
};

static const char *kNames[K_COUNT] = {
    "v_fma_f64", "v_mul_f64", "v_add_f64", "v_min_f64", "v_ldexp_f64", "v_rcp_f64", "v_cmp_gt_f64 (e64)",
    "v_fma_f64 sgpr operand", "v_fma_f64 inline const", "v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_and_b32",
    "v_lshl_add_u32", "v_ashrrev_i32", "v_cndmask_b32", "v_mov_b32", "v_mov_b64", "v_mov_b32 dpp quad_perm",
    "v_exp_f32", "v_rcp_f32", "v_cvt_f64_i32", "v_mul_lo_u32", "v_mad_u64_u32", "v_lshl_add_u64",
    "ds_read_b64", "ds_read_b128", "mix 1 fma_f64 + 1 add_u32 (per pair)", "mix 1 fma_f64 + 2 add_u32 (per triple)",
    "mix 1 fma_f64 + 1 cndmask (per pair)", "mix 1 fma_f64 + 1 ldexp_f64 (per pair)",
    "mix 4 fma_f64 + 1 ds_read_b128 (per five)",
    "sample-loop mix: 57 fma/mul/add_f64 + 6 min/max/ldexp_f64 + 1 rcp_f64 + 9 int32 + 3 ds_read_b128 (per 76)"};

template <int KIND>
__global__ void __launch_bounds__(1024) k_issue(unsigned long long *ticks, double *sink, int n, double seed)
{
    __shared__ double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6,
           a7 = seed + 7;
    const double b = 1.0000001, c = 1e-9;
    float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const float fb = 1.0000001f, fc = 1e-9f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    unsigned long long q0 = u0, q1 = u1, q2 = u2, q3 = u3, q4 = u4, q5 = u5, q6 = u6, q7 = u7;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w0 = {0, 0, 0, 0}, w1 = w0, w2 = w0, w3 = w0;
    const unsigned ldsaddr = (threadIdx.x & 127) * 16;
    const double sb = __builtin_amdgcn_readfirstlane((int)seed) + 1.0000001;   // lives in SGPRs
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#define A(k) a##k
#define F(k) f##k
#define U(k) u##k
#define Q(k) q##k
            if (KIND == K_FMA64) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A(k)) : "v"(b), "v"(c));
                REP8(X)
#undef X
            }
            if (KIND == K_MUL64) {
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(A(k)) : "v"(b));
                REP8(X)
#undef X
            }
            if (KIND == K_ADD64) {
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(A(k)) : "v"(c));
                REP8(X)
#undef X
            }
            if (KIND == K_MIN64) {
#define X(k) asm volatile("v_min_f64 %0, %0, %1" : "+v"(A(k)) : "v"(b));
                REP8(X)
#undef X
            }
            if (KIND == K_LDEXP64) {
#define X(k) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(A(k)) : "v"(i & 1));
                REP8(X)
#undef X
            }
            if (KIND == K_RCP64) {
#define X(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(A(k)));
                REP8(X)
#undef X
            }
            if (KIND == K_CMP64) {
#define X(k) asm volatile("v_cmp_gt_f64 vcc, %0, %1" ::"v"(A(k)), "v"(b) : "vcc");
                REP8(X)
#undef X
            }
            if (KIND == K_FMA64_SGPR) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A(k)) : "s"(sb), "v"(c));
                REP8(X)
#undef X
            }
            if (KIND == K_FMA64_LIT) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, 0.5" : "+v"(A(k)) : "v"(b));
                REP8(X)
#undef X
            }
            if (KIND == K_FMA32) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(F(k)) : "v"(fb), "v"(fc));
                REP8(X)
#undef X
            }
            if (KIND == K_PKFMA32) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(A(k)) : "v"(b), "v"(c));
                REP8(X)
#undef X
            }
            if (KIND == K_ADD_U32) {
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(U(k)) : "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_AND_B32) {
#define X(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(U(k)) : "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_LSHL_ADD_U32) {
#define X(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(U(k)) : "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_ASHR_I32) {
#define X(k) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(U(k)));
                REP8(X)
#undef X
            }
            if (KIND == K_CNDMASK) {
                // (no "vcc" clobber: with one the compiler puts an s_nop between every two)
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(U(k)) : "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_MOV_B32) {
#define X(k) asm volatile("v_mov_b32 %0, %1" : "+v"(U(k)) : "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_MOV_B64) {
#define X(k) asm volatile("v_mov_b64 %0, %1" : "+v"(A(k)) : "v"(b));
                REP8(X)
#undef X
            }
            if (KIND == K_MOV_DPP) {
#define X(k) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(U(k)));
                REP8(X)
#undef X
            }
            if (KIND == K_EXP32) {
#define X(k) asm volatile("v_exp_f32 %0, %0" : "+v"(F(k)));
                REP8(X)
#undef X
            }
            if (KIND == K_RCP32) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(F(k)));
                REP8(X)
#undef X
            }
            if (KIND == K_CVT_F64_I32) {
#define X(k) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(A(k)) : "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_MUL_LO_U32) {
#define X(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(U(k)) : "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_MAD_U64_U32) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(Q(k)) : "v"(i), "v"(u0) : "vcc");
                REP8(X)
#undef X
            }
            if (KIND == K_LSHL_ADD_U64) {
#define X(k) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(Q(k)) : "v"(q0));
                REP8(X)
#undef X
            }
            if (KIND == K_DS_READ_B64) {
#define X(k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(A(k)) : "v"(ldsaddr), "n"(k * 16));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (KIND == K_DS_READ_B128) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(w0) : "v"(ldsaddr));
                asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(w1) : "v"(ldsaddr));
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(w2) : "v"(ldsaddr));
                asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(w3) : "v"(ldsaddr));
                asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(w0) : "v"(ldsaddr));
                asm volatile("ds_read_b128 %0, %1 offset:2064" : "=v"(w1) : "v"(ldsaddr));
                asm volatile("ds_read_b128 %0, %1 offset:4112" : "=v"(w2) : "v"(ldsaddr));
                asm volatile("ds_read_b128 %0, %1 offset:6160" : "=v"(w3) : "v"(ldsaddr));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (KIND == K_MIX_FMA64_ADDU32) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_add_u32 %1, %1, %4" : "+v"(A(k)), "+v"(U(k)) : "v"(b), "v"(c), "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_MIX_FMA64_2ADDU32) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4" : "+v"(A(k)), "+v"(U(k)) : "v"(b), "v"(c), "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_MIX_FMA64_CNDMASK) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(A(k)), "+v"(U(k)) : "v"(b), "v"(c), "v"(i));
                REP8(X)
#undef X
            }
            if (KIND == K_MIX_FMA64_LDEXP) {
                asm volatile("v_fma_f64 %0, %0, %2, %3\n v_ldexp_f64 %1, %1, %4" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c), "v"(i & 1));
                asm volatile("v_fma_f64 %0, %0, %2, %3\n v_ldexp_f64 %1, %1, %4" : "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "v"(i & 1));
                asm volatile("v_fma_f64 %0, %0, %2, %3\n v_ldexp_f64 %1, %1, %4" : "+v"(a4), "+v"(a5) : "v"(b), "v"(c), "v"(i & 1));
                asm volatile("v_fma_f64 %0, %0, %2, %3\n v_ldexp_f64 %1, %1, %4" : "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(i & 1));
                asm volatile("v_fma_f64 %0, %0, %2, %3\n v_ldexp_f64 %1, %1, %4" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c), "v"(i & 1));
                asm volatile("v_fma_f64 %0, %0, %2, %3\n v_ldexp_f64 %1, %1, %4" : "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "v"(i & 1));
                asm volatile("v_fma_f64 %0, %0, %2, %3\n v_ldexp_f64 %1, %1, %4" : "+v"(a4), "+v"(a5) : "v"(b), "v"(c), "v"(i & 1));
                asm volatile("v_fma_f64 %0, %0, %2, %3\n v_ldexp_f64 %1, %1, %4" : "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(i & 1));
            }
            if (KIND == K_MIX_FMA64_DS128) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(w0) : "v"(ldsaddr));
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A(k)) : "v"(b), "v"(c));
                X(0) X(1) X(2) X(3)
                asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(w1) : "v"(ldsaddr));
                X(4) X(5) X(6) X(7)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (KIND == K_SAMPLE_MIX) {
                // the instruction multiset of one blackbody-side thick sample (tools/isa_hist.py),
                // as 8 independent chains so that nothing waits on a dependency: 76 instructions
                asm volatile("ds_read_b128 %0, %1" : "=v"(w0) : "v"(ldsaddr));
                asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(w1) : "v"(ldsaddr));
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(w2) : "v"(ldsaddr));
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A(k)) : "v"(b), "v"(c));
                REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) X(0)      // 57
#undef X
#define X(k) asm volatile("v_min_f64 %0, %0, %1" : "+v"(A(k)) : "v"(b));
                X(1) X(2) X(3)
#undef X
#define X(k) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(A(k)) : "v"(i & 1));
                X(4) X(5) X(6)
#undef X
                asm volatile("v_rcp_f64 %0, %0" : "+v"(a7));
#define X(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(U(k)) : "v"(i));
                X(0) X(1) X(2)
#undef X
#define X(k) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(U(k)));
                X(3) X(4) X(5)
#undef X
#define X(k) asm volatile("v_ashrrev_i32 %0, 7, %0" : "+v"(U(k)));
                X(6) X(7) X(0)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    asm volatile("" ::"v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
    asm volatile("" ::"v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7));
    asm volatile("" ::"v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7));
    asm volatile("" ::"v"(w0), "v"(w1), "v"(w2), "v"(w3));
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (sink && seed == -1234.5) sink[threadIdx.x] = a0 + f0 + u0 + (double)q0 + (double)w0.x;
}

static int per_iter(int kind)     // wave-instructions per pass of the j loop
{
    switch (kind) {
    case K_MIX_FMA64_ADDU32: case K_MIX_FMA64_CNDMASK: case K_MIX_FMA64_LDEXP: return 16;
    case K_MIX_FMA64_2ADDU32: return 24;
    case K_MIX_FMA64_DS128: return 10;
    case K_SAMPLE_MIX: return 76;
    default: return 8;
    }
}

template <int KIND>
static double run(int waves_per_simd, int grid, unsigned long long *d_ticks, int n)
{
    const int threads = 256 * waves_per_simd;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL((k_issue<KIND>), dim3(grid), dim3(threads), 0, 0, d_ticks, (double *)nullptr, n, 1.0);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d_ticks, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double instr_per_wave = (double)n * 4 * per_iter(KIND);
    return sum / grid / (instr_per_wave * waves_per_simd);
}

template <int KIND = 0>
static void sweep(unsigned long long *d_ticks, int grid, FILE *js, bool &first)
{
    if constexpr (KIND < K_COUNT) {
        const int ws[] = {1, 2, 3, 4};
        double r[4];
        for (int i = 0; i < 4; ++i) r[i] = run<KIND>(ws[i], grid, d_ticks, 400);
        printf("%-100s", kNames[KIND]);
        for (int i = 0; i < 4; ++i) printf("  W=%d: %6.2f", ws[i], r[i]);
        printf("\n");
        if (js) {
            fprintf(js, "%s\n  {\"class\": \"%s\", \"group\": %d, \"cycles_per_wave_instruction_per_simd\": {", first ? "" : ",", kNames[KIND], per_iter(KIND));
            for (int i = 0; i < 4; ++i) fprintf(js, "%s\"%d\": %.3f", i ? ", " : "", ws[i], r[i]);
            fprintf(js, "}}");
            first = false;
        }
        sweep<KIND + 1>(d_ticks, grid, js, first);
    }
}

int main(int argc, char **argv)
{
    int grid = 256;
    const char *json = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--grid") && i + 1 < argc) grid = atoi(argv[++i]);
        if (!strcmp(argv[i], "--json") && i + 1 < argc) json = argv[++i];
    }
    unsigned long long *d_ticks;
    hipMalloc(&d_ticks, 4096 * sizeof(unsigned long long));
    FILE *js = json ? fopen(json, "w") : nullptr;
    if (js) fprintf(js, "{\"unit\": \"shader cycles (s_memtime ticks) per wave64 instruction per SIMD; W = waves per SIMD\", \"grid_workgroups\": %d, \"rows\": [", grid);
    printf("cycles per wave-instruction per SIMD, %d workgroups (one per CU), workgroup = 4 W waves\n", grid);
    bool first = true;
    sweep<0>(d_ticks, grid, js, first);
    if (js) { fprintf(js, "\n]}\n"); fclose(js); }
    return 0;
}
