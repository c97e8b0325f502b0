// Micro-benchmark (development aid): issue cost of the vector instructions the pixel loops are made of, one wave's
// stream on one SIMD and eight waves per SIMD (the tile pass's small form): cycles per instruction from s_memtime
// around an unrolled stream of independent instructions.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void stream(unsigned long long *out, float seed) {
    v2f a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const v2f b = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ v4f lds[64];
    lds[threadIdx.x & 63] = (v4f){seed, seed, seed, seed};
    v4f q4 = lds[0];
    const unsigned zero = 0u;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        if (KIND == 0) {  // v_fma_f32 x 16
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 1) {  // v_pk_fma_f32 x 16
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 2) {  // v_exp_f32 x 16
            asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 3) {  // v_pk_mul_f32 x 16
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 4) {  // v_cndmask x 16
            asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "vcc");
        } else if (KIND == 5) {  // 8 independent v_pk_fma on 8 different registers, twice (no back-to-back dependency)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 6) {  // the same with v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 7) {  // exp on 8 different registers, twice
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 8) {  // v_cndmask_b32_e64 with an SGPR-pair mask
            asm volatile("s_mov_b64 s[20:21], 0x5555aaaa\n" REP8("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "s20", "s21");
        } else if (KIND == 9) {  // v_cmp_gt_u32_e64 into SGPR pairs + v_cndmask reading them (the pixel loops' pair)
            asm volatile(REP8("v_cmp_gt_u32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %1, 0, %1, s[20:21]\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "s20", "s21");
        } else if (KIND == 10) {  // v_mov_b32
            asm volatile(REP8("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 11) {  // v_add_u32
            asm volatile(REP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 12) {  // v_pk_add_f32
            asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 13) {  // v_bfi_b32 + v_ashrrev_i32 (the integer form of the pair test)
            asm volatile(REP8("v_bfi_b32 %0, %0, 0, %8\n v_ashrrev_i32 %1, 31, %1\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 14) {  // v_cmp_gt_u32_e32 (vcc) alone
            asm volatile(REP8("v_cmp_gt_u32_e32 vcc, %0, %8\n v_cmp_gt_u32_e32 vcc, %1, %8\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "vcc");
        } else if (KIND == 15) {  // v_mul_f32
            asm volatile(REP8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 16) {  // ds_read_b128 broadcast x 16 (address 0), waited once
            asm volatile(REP8("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:16\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(q4) : "v"(zero));
        } else if (KIND == 17) {  // v_cmp_gt_u32_e32 (vcc) + v_cndmask_b32_e32 (vcc): the pair as the compiler often emits it
            asm volatile(REP8("v_cmp_gt_u32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, 0, %1, vcc\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "vcc");
        } else if (KIND == 18) {  // v_cndmask_b32_e64 with vcc as its mask operand
            asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "vcc");
        } else if (KIND == 19) {  // v_cndmask_b32_e32 (vcc), eight independent registers
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "vcc");
        } else if (KIND == 20) {  // v_cndmask_b32_e32 with a constant 0 source, as in the loops: v_cndmask_b32_e32 d, 0, s, vcc
            asm volatile(REP8("v_cndmask_b32_e32 %0, 0, %8, vcc\n v_cndmask_b32_e32 %1, 0, %9, vcc\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "vcc");
        } else if (KIND == 21) {  // the item loop's pattern: v_cmp -> vcc, s_and_b64 vcc, <lane mask>, vcc, v_cndmask_b32_e32 (vcc): 3 instructions x 8
            asm volatile("s_mov_b64 s[20:21], -1\n" REP8("v_cmp_gt_u32_e32 vcc, %0, %8\n s_and_b64 vcc, s[20:21], vcc\n v_cndmask_b32_e32 %1, 0, %1, vcc\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "vcc", "scc", "s20", "s21");
        } else if (KIND == 22) {  // the same through an SGPR pair: v_cmp_e64 -> s[22:23], s_and_b64 s[22:23], v_cndmask_b32_e64
            asm volatile("s_mov_b64 s[20:21], -1\n" REP8("v_cmp_gt_u32_e64 s[22:23], %0, %8\n s_and_b64 s[22:23], s[20:21], s[22:23]\n v_cndmask_b32_e64 %1, 0, %1, s[22:23]\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "scc", "s22", "s23", "s20", "s21");
        } else if (KIND == 23) {  // no scalar AND: the lane mask folded into the compared value beforehand (v_cmp + v_cndmask only, 2 x 8)
            asm volatile(REP8("v_cmp_gt_u32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, 0, %1, vcc\n") REP8("v_mov_b32 %2, %3\n") : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x) : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y + q4.x == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND>
static void run(const char *name, unsigned long long *out) {
    for (int waves_per_simd : {1, 2, 4, 8}) {
        // one workgroup of 256 lanes = one wave per SIMD of a CU; 256 CUs; w workgroups per CU
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((stream<KIND>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream<KIND>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long cyc;
        CHECK(hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost));
        // s_memtime ticks at 100 MHz: convert with the wall time instead -- instructions per SIMD / time
        const double insts_per_simd = 256.0 * 16 * waves_per_simd;
        printf("%-28s %d waves/SIMD: %.3f us total, %.2f ns per instruction per SIMD (= %.2f cycles at 2.4 GHz), counter %llu\n", name,
               waves_per_simd, ms * 1e3, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4, cyc);
    }
}

int main() {
    unsigned long long *out;
    CHECK(hipMalloc(&out, 64));
    run<0>("v_fma_f32 (dependent pairs)", out);
    run<6>("v_fma_f32 (independent)", out);
    run<1>("v_pk_fma_f32 (dep. pairs)", out);
    run<5>("v_pk_fma_f32 (independent)", out);
    run<3>("v_pk_mul_f32", out);
    run<2>("v_exp_f32 (dep. pairs)", out);
    run<7>("v_exp_f32 (independent)", out);
    run<4>("v_cndmask_b32 (vcc)", out);
    run<19>("v_cndmask_b32 (vcc) indep.", out);
    run<20>("v_cndmask_b32 d,0,s,vcc", out);
    run<18>("v_cndmask_b32_e64 (vcc)", out);
    run<17>("v_cmp_e32 + v_cndmask_e32", out);
    run<21>("cmp; s_and vcc; cndmask e32 (x24)", out);
    run<22>("cmp64; s_and sgpr; cndmask e64 (x24)", out);
    run<23>("cmp; cndmask e32 x8 + 8 v_mov (x24)", out);
    run<8>("v_cndmask_b32_e64 (sgpr)", out);
    run<9>("v_cmp_e64 + v_cndmask_e64", out);
    run<14>("v_cmp_gt_u32_e32 (vcc)", out);
    run<10>("v_mov_b32", out);
    run<11>("v_add_u32", out);
    run<15>("v_mul_f32", out);
    run<12>("v_pk_add_f32", out);
    run<13>("v_bfi_b32 + v_ashrrev_i32", out);
    run<16>("ds_read_b128 (broadcast)", out);
    return 0;
}
