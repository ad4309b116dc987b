// ubench_isa.hip — per-instruction issue cost on gfx950 for the integer ops the 256-bit
// Montgomery product is built from (inline asm so that hipcc cannot fold or reorder them).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t a = threadIdx.x + 1, b = blockIdx.x * 3 + 7;
    uint64_t c0 = a, c1 = b, c2 = a + 5, c3 = b + 9, c4 = 3, c5 = 4, c6 = 5, c7 = 6;
    uint32_t d0 = a, d1 = b, d2 = 3, d3 = 4, d4 = 5, d5 = 6, d6 = 7, d7 = 8;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) {  // 8 independent v_mad_u64_u32 per group, 8 groups
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 1) {  // dependent chain of v_mad_u64_u32
            REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n" : "+v"(c0) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 2) {  // v_add_co_u32 / v_addc_co_u32 independent
            REP8(asm volatile("v_addc_co_u32 %0, vcc, %8, %0, vcc\n v_addc_co_u32 %1, vcc, %8, %1, vcc\n v_addc_co_u32 %2, vcc, %8, %2, vcc\n v_addc_co_u32 %3, vcc, %8, %3, vcc\n"
                              "v_addc_co_u32 %4, vcc, %8, %4, vcc\n v_addc_co_u32 %5, vcc, %8, %5, vcc\n v_addc_co_u32 %6, vcc, %8, %6, vcc\n v_addc_co_u32 %7, vcc, %8, %7, vcc\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a) : "vcc");)
        } else if (OP == 3) {  // v_lshl_add_u64
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                              "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(c0));)
        } else if (OP == 4) {  // v_mov_b32
            REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a));)
        } else if (OP == 5) {  // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a));)
        } else if (OP == 6) {  // v_mul_hi_u32
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a));)
        } else if (OP == 7) {  // v_add_u32 (plain)
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a));)
        } else if (OP == 8) {  // v_fma_f64
            double e0 = __longlong_as_double(c0), e1 = __longlong_as_double(c1), e2 = __longlong_as_double(c2), e3 = __longlong_as_double(c3);
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(1.0000001), "v"(0.5));)
            c0 = __double_as_longlong(e0); c1 = __double_as_longlong(e1); c2 = __double_as_longlong(e2); c3 = __double_as_longlong(e3);
        } else if (OP == 9) {  // v_lshrrev_b64
            REP8(asm volatile("v_lshrrev_b64 %0, 29, %0\n v_lshrrev_b64 %1, 29, %1\n v_lshrrev_b64 %2, 29, %2\n v_lshrrev_b64 %3, 29, %3\n v_lshrrev_b64 %4, 29, %4\n v_lshrrev_b64 %5, 29, %5\n v_lshrrev_b64 %6, 29, %6\n v_lshrrev_b64 %7, 29, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 10) {  // v_alignbit_b32
            REP8(asm volatile("v_alignbit_b32 %0, %8, %0, 29\n v_alignbit_b32 %1, %8, %1, 29\n v_alignbit_b32 %2, %8, %2, 29\n v_alignbit_b32 %3, %8, %3, 29\n v_alignbit_b32 %4, %8, %4, 29\n v_alignbit_b32 %5, %8, %5, 29\n v_alignbit_b32 %6, %8, %6, 29\n v_alignbit_b32 %7, %8, %7, 29\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 11) {  // v_and_b32 (literal)
            REP8(asm volatile("v_and_b32 %0, 0x1fffffff, %0\n v_and_b32 %1, 0x1fffffff, %1\n v_and_b32 %2, 0x1fffffff, %2\n v_and_b32 %3, 0x1fffffff, %3\n v_and_b32 %4, 0x1fffffff, %4\n v_and_b32 %5, 0x1fffffff, %5\n v_and_b32 %6, 0x1fffffff, %6\n v_and_b32 %7, 0x1fffffff, %7\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 12) {  // v_mov_b64
            REP8(asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(c0), "v"(b) : "vcc");)
        } else if (OP == 13) {  // v_add3_u32
            REP8(asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 14) {  // v_mad_u32_u24
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 15) {  // v_bfe_u32
            REP8(asm volatile("v_bfe_u32 %0, %0, 3, 29\n v_bfe_u32 %1, %1, 3, 29\n v_bfe_u32 %2, %2, 3, 29\n v_bfe_u32 %3, %3, 3, 29\n v_bfe_u32 %4, %4, 3, 29\n v_bfe_u32 %5, %5, 3, 29\n v_bfe_u32 %6, %6, 3, 29\n v_bfe_u32 %7, %7, 3, 29\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 16) {  // v_lshl_add_u32
            REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 17) {  // v_mad_u64_u32 (sgpr src)
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "s"(iters) : "vcc");)
        } else if (OP == 18) {  // v_and_or_b32
            REP8(asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 19) {  // v_sub_u32
            REP8(asm volatile("v_sub_u32 %0, %8, %0\n v_sub_u32 %1, %8, %1\n v_sub_u32 %2, %8, %2\n v_sub_u32 %3, %8, %3\n v_sub_u32 %4, %8, %4\n v_sub_u32 %5, %8, %5\n v_sub_u32 %6, %8, %6\n v_sub_u32 %7, %8, %7\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 20) {  // v_lshrrev_b32
            REP8(asm volatile("v_lshrrev_b32 %0, 29, %0\n v_lshrrev_b32 %1, 29, %1\n v_lshrrev_b32 %2, 29, %2\n v_lshrrev_b32 %3, 29, %3\n v_lshrrev_b32 %4, 29, %4\n v_lshrrev_b32 %5, 29, %5\n v_lshrrev_b32 %6, 29, %6\n v_lshrrev_b32 %7, 29, %7\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 21) {  // v_mul_u32_u24
            REP8(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 22) {  // v_mul_hi_u32_u24
            REP8(asm volatile("v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 23) {  // v_cndmask_b32
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 24) {  // 4 v_mad_u64_u32 + 4 v_and_b32 interleaved: 8 instructions per statement
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_and_b32 %4, 0x1fffffff, %4\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_and_b32 %5, 0x1fffffff, %5\n"
                              "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_and_b32 %6, 0x1fffffff, %6\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_and_b32 %7, 0x1fffffff, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 26) {  // 4 v_mad_u64_u32, then a RUN of 4 v_and_b32 (do plain instructions pair up when they are adjacent?)
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_and_b32 %4, 0x1fffffff, %4\n v_and_b32 %5, 0x1fffffff, %5\n v_and_b32 %6, 0x1fffffff, %6\n v_and_b32 %7, 0x1fffffff, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 27) {  // 8 mads, then a run of 8 v_and_b32
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_and_b32 %4, 0x1fffffff, %4\n v_and_b32 %5, 0x1fffffff, %5\n v_and_b32 %6, 0x1fffffff, %6\n v_and_b32 %7, 0x1fffffff, %7\n"
                              "v_and_b32 %4, 0x1ffffffe, %4\n v_and_b32 %5, 0x1ffffffe, %5\n v_and_b32 %6, 0x1ffffffe, %6\n v_and_b32 %7, 0x1ffffffe, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");)
        } else if (OP == 25) {  // 4 v_mad_u64_u32 + 4 v_lshrrev_b64 interleaved
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_lshrrev_b64 %4, 29, %4\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_lshrrev_b64 %5, 29, %5\n"
                              "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_lshrrev_b64 %6, 29, %6\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_lshrrev_b64 %7, 29, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7) + d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}

template <int OP>
void run(const char* name, uint32_t* out, int waves_per_simd) {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * waves_per_simd, iters = 2000;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    // each wave issues iters*64 instructions; waves_per_simd waves share a SIMD
    const double cyc = ms * 1e-3 * (prop.clockRate * 1e3) / ((double)iters * 64 * waves_per_simd);
    printf("%-28s %d waves/SIMD: %7.3f ms  -> %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms, cyc);
}

int main() {
    uint32_t* out; CHK(hipMalloc(&out, 256 * 256 * 16 * 4));
    for (int w : {1, 4}) {
        run<0>("v_mad_u64_u32 (indep)", out, w);
        run<1>("v_mad_u64_u32 (dep chain)", out, w);
        run<2>("v_addc_co_u32", out, w);
        run<3>("v_lshl_add_u64", out, w);
        run<4>("v_mov_b32", out, w);
        run<5>("v_mul_lo_u32", out, w);
        run<6>("v_mul_hi_u32", out, w);
        run<7>("v_add_u32", out, w);
        run<8>("v_fma_f64", out, w);
        run<9>("v_lshrrev_b64", out, w);
        run<10>("v_alignbit_b32", out, w);
        run<11>("v_and_b32 (literal)", out, w);
        run<12>("v_mov_b64", out, w);
        run<13>("v_add3_u32", out, w);
        run<14>("v_mad_u32_u24", out, w);
        run<15>("v_bfe_u32", out, w);
        run<16>("v_lshl_add_u32", out, w);
        run<17>("v_mad_u64_u32 (sgpr src)", out, w);
        run<18>("v_and_or_b32", out, w);
        run<19>("v_sub_u32", out, w);
        run<20>("v_lshrrev_b32", out, w);
        run<21>("v_mul_u32_u24", out, w);
        run<22>("v_mul_hi_u32_u24", out, w);
        run<23>("v_cndmask_b32", out, w);
        run<24>("4 mad + 4 v_and (per 8 instr)", out, w);
        run<25>("4 mad + 4 v_lshrrev_b64", out, w);
        run<26>("4 mad, then 4 v_and (per 8 instr)", out, w);
        run<27>("8 mad, then 8 v_and (per 16 instr)", out, w);
    }
    return 0;
}
