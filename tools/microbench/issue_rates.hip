// issue_rates.hip -- measures per-SIMD issue cost (cycles per wave64 instruction) of the instruction kinds the
// compositing loops are made of, on the GPU it runs on.  Tuning aid (DESIGN.md 4 quotes its output); not part of
// the product.  Build: hipcc -O2 --offload-arch=gfx950 issue_rates.hip -o issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kIters = 2048;

#define REP8(x) x x x x x x x x

enum Op { DPPADD, DPPBC, SWAP32, SWAP16, DSADD1, DSADD64, WRLANE, BPERM, DIVF, CND64, CNDNEW, CNDZERO, ADD, SUB, ANDB, CMP64, RFL, MED3, MAX, BRANCH, LDSTP, CMPX, MUL,  PKMUL, FMA, EXP, CNDMASK, CMP, MULDEP, SALU, SALUDEP, LDSDEP, MIX_VS, PKADD, MOV, MIN, RCP, FFS_CHAIN };

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, float seed)
{
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float m = 1.0000001f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a0, a3}, p5 = {a1, a4}, p6 = {a2, a5}, p7 = {a6, a1};
    v2f pm = {m, m};
    unsigned long long s0 = 0x123456789abcdefull, s1 = 0xfedcba987654321ull;
    uint32_t addr = 0;
    for (int i = 0; i < kIters; ++i) {
        if (OP == MUL) {
            asm volatile(REP8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (OP == FMA) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (OP == PKMUL) {
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
        } else if (OP == PKADD) {
            asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
        } else if (OP == EXP) {
            asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == RCP) {
            asm volatile(REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == MOV) {
            asm volatile(REP8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == MIN) {
            asm volatile(REP8("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                              "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (OP == CNDMASK) {
            asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
        } else if (OP == CMP) {
            asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
        } else if (OP == CND64) {
            asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                              "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "s20", "s21");
        } else if (OP == CNDNEW) {
            // fresh mask per select, as in real code: compare, then select on its result
            asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
        } else if (OP == CNDZERO) {
            asm volatile(REP8("v_cndmask_b32 %0, 0, %8, vcc\n v_cndmask_b32 %1, 0, %8, vcc\n v_cndmask_b32 %2, 0, %8, vcc\n v_cndmask_b32 %3, 0, %8, vcc\n"
                              "v_cndmask_b32 %4, 0, %8, vcc\n v_cndmask_b32 %5, 0, %8, vcc\n v_cndmask_b32 %6, 0, %8, vcc\n v_cndmask_b32 %7, 0, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
        } else if (OP == ADD) {
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (OP == SUB) {
            asm volatile(REP8("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                              "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (OP == ANDB) {
            asm volatile(REP8("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                              "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (OP == MAX) {
            asm volatile(REP8("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                              "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (OP == MED3) {
            asm volatile(REP8("v_med3_f32 %0, %0, %8, %8\n v_med3_f32 %1, %1, %8, %8\n v_med3_f32 %2, %2, %8, %8\n v_med3_f32 %3, %3, %8, %8\n"
                              "v_med3_f32 %4, %4, %8, %8\n v_med3_f32 %5, %5, %8, %8\n v_med3_f32 %6, %6, %8, %8\n v_med3_f32 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (OP == CMP64) {
            asm volatile(REP8("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[20:21], %2, %8\n v_cmp_lt_f32 s[22:23], %3, %8\n"
                              "v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[20:21], %6, %8\n v_cmp_lt_f32 s[22:23], %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "s20", "s21", "s22", "s23");
        } else if (OP == RFL) {
            asm volatile(REP8("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s21, %3\n"
                              "v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s21, %5\n v_readfirstlane_b32 s20, %6\n v_readfirstlane_b32 s21, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "s20", "s21");
        } else if (OP == BRANCH) {
            // 8 taken forward branches (each skips one s_nop) per REP8 element: the cost of a taken scalar branch
            asm volatile(REP8("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1\n s_nop 0\n s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1\n s_nop 0\n s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1\n s_nop 0\n s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1\n s_nop 0\n")
                         : : : "scc");
        } else if (OP == LDSTP) {
            // independent broadcast reads: LDS issue throughput
            asm volatile(REP8("ds_read_b128 v[20:23], %0\n ds_read_b128 v[24:27], %0 offset:16\n ds_read_b128 v[28:31], %0 offset:32\n ds_read_b128 v[32:35], %0 offset:48\n") "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(addr) : : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "memory");
        } else if (OP == CMPX) {
            // compare + scalar and + select: the usual predicate pattern
            asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_gt_f32 s[20:21], %1, %8\n s_and_b64 vcc, vcc, s[20:21]\n v_cndmask_b32 %2, %2, %8, vcc\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_gt_f32 s[20:21], %5, %8\n s_and_b64 vcc, vcc, s[20:21]\n v_cndmask_b32 %6, %6, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc", "s20", "s21", "scc");
        } else if (OP == DPPADD) {
            asm volatile(REP8("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == DPPBC) {
            asm volatile(REP8("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == SWAP32) {
            asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == SWAP16) {
            asm volatile(REP8("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                              "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == DSADD1) {
            // LDS float add from one active lane (the rest masked off by exec), as the gradient flush does
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 1\n"
                         REP8("ds_add_f32 %0, %1\n ds_add_f32 %0, %2 offset:1024\n ds_add_f32 %0, %3 offset:2048\n ds_add_f32 %0, %4 offset:3072\n")
                         "s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, s[20:21]\n"
                         : : "v"(addr), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "memory");
        } else if (OP == DSADD64) {
            // all 64 lanes, distinct consecutive addresses
            asm volatile(REP8("ds_add_f32 %0, %1\n ds_add_f32 %0, %2 offset:1024\n ds_add_f32 %0, %3 offset:2048\n ds_add_f32 %0, %4 offset:3072\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : : "v"((threadIdx.x & 63u) * 4u), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        } else if (OP == WRLANE) {
            asm volatile(REP8("v_writelane_b32 %0, s20, 15\n v_writelane_b32 %1, s20, 31\n v_writelane_b32 %2, s20, 47\n v_writelane_b32 %3, s20, 63\n"
                              "v_writelane_b32 %4, s20, 15\n v_writelane_b32 %5, s20, 31\n v_writelane_b32 %6, s20, 47\n v_writelane_b32 %7, s20, 63\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20");
        } else if (OP == BPERM) {
            asm volatile(REP8("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                              "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"((threadIdx.x & 63u) * 4u ^ 128u) : "memory");
        } else if (OP == DIVF) {
            // IEEE-correct float division as hipcc emits it for a / b (8 per iteration)
            a0 = a0 / m; a1 = a1 / m; a2 = a2 / m; a3 = a3 / m; a4 = a4 / m; a5 = a5 / m; a6 = a6 / m; a7 = a7 / m;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(m));
        } else if (OP == MULDEP) {
            asm volatile(REP8("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                              "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n")
                         : "+v"(a0) : "v"(m));
        } else if (OP == SALU) {
            asm volatile(REP8("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                              "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                         : "+s"(((uint32_t*)&s0)[0]), "+s"(((uint32_t*)&s0)[1]), "+s"(((uint32_t*)&s1)[0]), "+s"(((uint32_t*)&s1)[1]) : : "scc");
        } else if (OP == SALUDEP) {
            asm volatile(REP8("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                              "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n")
                         : "+s"(((uint32_t*)&s0)[0]) : : "scc");
        } else if (OP == FFS_CHAIN) {
            // the scalar skeleton of one compositing step: find-first-set, clear the bit, form an address
            asm volatile(REP8("s_ff1_i32_b32 s20, %0\n s_add_u32 s22, %0, -1\n s_and_b32 %0, %0, s22\n s_lshl_b32 s21, s20, 4\n s_or_b32 %0, %0, s21\n")
                         : "+s"(((uint32_t*)&s0)[0]) : : "scc", "s20", "s21", "s22");
        } else if (OP == LDSDEP) {
            // broadcast read whose address depends on the previous read (0 is stored everywhere): pure LDS latency
            asm volatile(REP8("ds_read_b128 v[20:23], %0\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v20\n")
                         : "+v"(addr) : : "v20", "v21", "v22", "v23", "memory");
        } else if (OP == MIX_VS) {
            asm volatile(REP8("v_mul_f32 %0, %0, %8\n s_add_u32 %9, %9, 1\n v_mul_f32 %1, %1, %8\n s_add_u32 %10, %10, 1\n v_mul_f32 %2, %2, %8\n s_add_u32 %9, %9, 1\n v_mul_f32 %3, %3, %8\n s_add_u32 %10, %10, 1\n"
                              "v_mul_f32 %4, %4, %8\n s_add_u32 %9, %9, 1\n v_mul_f32 %5, %5, %8\n s_add_u32 %10, %10, 1\n v_mul_f32 %6, %6, %8\n s_add_u32 %9, %9, 1\n v_mul_f32 %7, %7, %8\n s_add_u32 %10, %10, 1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(m), "s"(((uint32_t*)&s0)[0]), "s"(((uint32_t*)&s1)[0]) : "scc");
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(s0 + s1) + (float)addr;
    if (r == 12345.678f) out[0] = r;
}

template <int OP>
void run(const char* name, int instr_per_iter, float* d_out, int waves_per_simd, double clock_ghz, int cus)
{
    // one workgroup of 256 threads = one wave per SIMD of a CU
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d_out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d_out, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)kIters * instr_per_iter * waves_per_simd;
    const double cycles         = ms * 1e-3 * clock_ghz * 1e9;
    printf("%-10s waves/SIMD=%d  %8.3f ms  %6.2f cycles per instruction per SIMD (at %.2f GHz)\n", name, waves_per_simd, ms,
           cycles / instr_per_simd, clock_ghz);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate * 1e-6;
    const int    cus = prop.multiProcessorCount;
    printf("%s: %d CUs, clockRate %.3f GHz\n", prop.name, cus, ghz);
    float* d_out;
    CHECK(hipMalloc(&d_out, 4));
    for (int w : {1, 4, 8}) {
        run<DPPADD>("add_dpp shr", 64, d_out, w, ghz, cus);
        run<DPPBC>("add_dpp bcast", 64, d_out, w, ghz, cus);
        run<SWAP32>("permlane32sw", 64, d_out, w, ghz, cus);
        run<SWAP16>("permlane16sw", 64, d_out, w, ghz, cus);
        run<DSADD1>("ds_add 1lane", 32, d_out, w, ghz, cus);
        run<DSADD64>("ds_add 64ln", 32, d_out, w, ghz, cus);
        run<WRLANE>("v_writelane", 64, d_out, w, ghz, cus);
        run<BPERM>("ds_bpermute", 64, d_out, w, ghz, cus);
        run<DIVF>("ieee div", 8, d_out, w, ghz, cus);
        run<ADD>("v_add", 64, d_out, w, ghz, cus);
        run<SUB>("v_sub", 64, d_out, w, ghz, cus);
        run<ANDB>("v_and", 64, d_out, w, ghz, cus);
        run<MAX>("v_max", 64, d_out, w, ghz, cus);
        run<MED3>("v_med3", 64, d_out, w, ghz, cus);
        run<CND64>("cnd sgpr", 64, d_out, w, ghz, cus);
        run<CNDNEW>("cmp+cnd", 64, d_out, w, ghz, cus);
        run<CNDZERO>("cnd 0", 64, d_out, w, ghz, cus);
        run<CMP64>("cmp sgpr", 64, d_out, w, ghz, cus);
        run<CMPX>("cmpcmpandcnd", 64, d_out, w, ghz, cus);
        run<RFL>("readfirstl", 64, d_out, w, ghz, cus);
        run<BRANCH>("cmp+branch", 64, d_out, w, ghz, cus);
        run<LDSTP>("lds b128", 32, d_out, w, ghz, cus);
        run<MUL>("v_mul", 64, d_out, w, ghz, cus);
        run<FMA>("v_fma", 64, d_out, w, ghz, cus);
        run<PKMUL>("v_pk_mul", 64, d_out, w, ghz, cus);
        run<PKADD>("v_pk_add", 64, d_out, w, ghz, cus);
        run<EXP>("v_exp", 64, d_out, w, ghz, cus);
        run<RCP>("v_rcp", 64, d_out, w, ghz, cus);
        run<MOV>("v_mov", 64, d_out, w, ghz, cus);
        run<MIN>("v_min", 64, d_out, w, ghz, cus);
        run<CNDMASK>("v_cndmask", 64, d_out, w, ghz, cus);
        run<CMP>("v_cmp", 64, d_out, w, ghz, cus);
        run<MULDEP>("v_mul dep", 64, d_out, w, ghz, cus);
        run<SALU>("s_add", 64, d_out, w, ghz, cus);
        run<SALUDEP>("s_add dep", 64, d_out, w, ghz, cus);
        run<FFS_CHAIN>("ffs chain", 40, d_out, w, ghz, cus);
        run<LDSDEP>("lds dep", 8, d_out, w, ghz, cus);
        run<MIX_VS>("v+s mix", 128, d_out, w, ghz, cus);
    }
    return 0;
}
