// Dev tool: how many cycles a wave64 VALU instruction holds its SIMD on gfx950 (the denominator of "VALU busy"):
// independent chains of v_add_u32 / v_fma_f32 / v_lshlrev_b64 / v_cndmask, timed with s_memtime, 1 / 2 / 4 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o uncalled_amd/variants/ubench_valu tools/dev/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
constexpr int ITER = 2000;
template <int OP> __global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) {
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
        } else if (OP == 1) {
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(f0));
        } else if (OP == 2) {
            asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n"
                         "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        } else if (OP == 3) {
            asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_u32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         "v_cmp_gt_u32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_gt_u32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if (OP == 4) {   // dependent chain
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                         "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" : "+v"(a0) : "v"(seed));
        } else if (OP == 5) {   // 64-bit compare + mbcnt pair (ballot idiom)
            asm volatile("v_cmp_gt_u64 vcc, %0, %1\n v_mbcnt_lo_u32_b32 %2, vcc_lo, 0\n v_mbcnt_hi_u32_b32 %2, vcc_hi, %2\n v_add_u32 %3, %3, %2\n"
                         "v_cmp_gt_u64 vcc, %1, %0\n v_mbcnt_lo_u32_b32 %4, vcc_lo, 0\n v_mbcnt_hi_u32_b32 %4, vcc_hi, %4\n v_add_u32 %5, %5, %4\n"
                         : "+v"(q0), "+v"(q1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : : "vcc");
        } else if (OP == 6) {   // DPP moves
            asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %5, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3);
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = (t1 - t0) | ((uint64_t)(s & 1) << 63);
}
template <int OP> void run(const char *name, uint64_t *d, int threads) {
    k<OP><<<256, threads>>>(d, 7); CHECK(hipDeviceSynchronize());
    k<OP><<<256, threads>>>(d, 7); CHECK(hipDeviceSynchronize());
    uint64_t h[4]; CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    double cyc = (double)(h[0] & ~(1ull << 63));
    printf("%-28s %2d waves/SIMD: %.2f cycles (s_memtime) per wave-instruction per wave = %.2f SIMD-cycles per instruction\n", name, threads / 256,
           cyc / (ITER * 8.0), cyc / (ITER * 8.0) / (threads / 256));
}
template <int OP> void run_all(const char *name, uint64_t *d) { run<OP>(name, d, 256); run<OP>(name, d, 512); run<OP>(name, d, 1024); }
int main() {
    uint64_t *d; CHECK(hipMalloc(&d, 256 * 4 * 8 * 4));
    run_all<0>("v_add_u32 x8 independent", d); run_all<1>("v_fma_f32 x8 independent", d); run_all<2>("v_lshlrev_b64 x8 (4 chains)", d);
    run_all<3>("v_cmp+v_cndmask x4", d); run_all<4>("v_add_u32 dependent chain", d); run_all<5>("v_cmp_u64+mbcnt pair+add", d); run_all<6>("v_mov_b32_dpp x8", d);
    return 0;
}
