// Dev tool: what one vector-memory instruction costs a CU when every wavefront of the chip is issuing them, by access shape.
// k_map is bound by the per-CU vector memory pipeline (round 3): this measures the price list it is optimised against.
//   hipcc --offload-arch=gfx950 -O3 -o uncalled_amd/variants/ubench_vmem tools/dev/ubench_vmem.hip ; run on the GPU box
// Each wavefront (one per workgroup, 12 per CU as in k_map) owns a private region of REGION bytes (far beyond L2 in total)
// and issues ITER x UNROLL instructions of one shape on it; reported: ns per wave-instruction per CU-resident wave and the
// implied cycles of the shared pipeline per instruction (12 waves share a CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef char *gptr;

constexpr int WAVES_PER_CU = 12;
constexpr uint32_t REGION = 160 * 1024;     // bytes per wave: k_map's per-event working set
constexpr int ITER = 400;

// shape: 0 = 16 B per lane, lanes 64 B apart at random 64-B records (gather of one quarter of a record)
//        1 = 16 B per lane, contiguous (1 KB per instruction)
//        2 = 4 x 16 B per lane covering one random 64-B record (four instructions)
//        3 = 8 B per lane at random 8-B slots
//        4 = 4 B per lane at random 4-B slots
//        5 = 8 B per lane contiguous
//        6 = 16 B per lane, 4 adjacent lanes cover one random 64-B record (one instruction = 16 records)
//        7 = 16 B per lane, lanes 64 B apart, records consecutive (lane l -> record base + l)
//        8 = 8 B per lane, lane l writes/reads 12 consecutive keys at stride 96 B (merge output shape), one of the 12
template <int SHAPE, bool STORE>
__global__ __launch_bounds__(64, 3) void k_bench(char *base_, uint32_t *sink, uint32_t seed) {
    const gptr base = (gptr)base_ + (size_t)blockIdx.x * REGION;
    const uint32_t lane = threadIdx.x;
    uint32_t x = seed + blockIdx.x * 977u + lane * 131u;
    uint32_t acc = 0;
    const uint32_t nrec = REGION / 64;
    for (int it = 0; it < ITER; ++it) {
        x = x * 1664525u + 1013904223u;
        const uint32_t r = (x >> 8) % nrec;
        uint32_t off;
        if (SHAPE == 0) off = r * 64 + ((x >> 4) & 3u) * 16;
        else if (SHAPE == 1) off = (((uint32_t)it * 1024u) % (REGION - 1024)) + lane * 16;
        else if (SHAPE == 2) off = r * 64;
        else if (SHAPE == 3) off = ((x >> 8) % (REGION / 8)) * 8;
        else if (SHAPE == 4) off = ((x >> 8) % (REGION / 4)) * 4;
        else if (SHAPE == 5) off = (((uint32_t)it * 512u) % (REGION - 512)) + lane * 8;
        else if (SHAPE == 6) { const uint32_t rq = (uint32_t)__shfl((int)r, (int)(lane & ~3u)); off = rq * 64 + (lane & 3u) * 16; }
        else if (SHAPE == 7) { const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r) % (nrec - 64); off = (r0 + lane) * 64 + ((it & 3) * 16); }
        else { const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)x) % ((REGION - 64 * 96) / 8); off = r0 * 8 + lane * 96 + (it % 12) * 8; }
        if (STORE) {
            if (SHAPE == 2) {
                const uint4 v = make_uint4(x, acc, lane, (uint32_t)it);
                *(uint4 *)(base + off) = v; *(uint4 *)(base + off + 16) = v;
                *(uint4 *)(base + off + 32) = v; *(uint4 *)(base + off + 48) = v;
            } else if (SHAPE == 3 || SHAPE == 5 || SHAPE == 8) *(uint2 *)(base + off) = make_uint2(x, lane);
            else if (SHAPE == 4) *(uint32_t *)(base + off) = x;
            else *(uint4 *)(base + off) = make_uint4(x, acc, lane, (uint32_t)it);
        } else {
            if (SHAPE == 2) {
                const uint4 a = *(uint4 *)(base + off), b = *(uint4 *)(base + off + 16),
                            c = *(uint4 *)(base + off + 32), d = *(uint4 *)(base + off + 48);
                acc += a.x ^ b.y ^ c.z ^ d.w;
            } else if (SHAPE == 3 || SHAPE == 5 || SHAPE == 8) { const uint2 a = *(uint2 *)(base + off); acc += a.x ^ a.y; }
            else if (SHAPE == 4) acc += *(uint32_t *)(base + off);
            else { const uint4 a = *(uint4 *)(base + off); acc += a.x ^ a.w; }
            // eight loads in flight per wave, as a software-pipelined phase of k_map has: the dependence on acc is only through the sum
            if ((it & 7) == 7) x ^= acc & 1u;
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int SHAPE, bool STORE> static void run(const char *name, char *buf, uint32_t *sink, int n_waves, int n_cu, double clk_ghz) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_bench<SHAPE, STORE>), dim3(n_waves), dim3(64), 0, 0, buf, sink, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_bench<SHAPE, STORE>), dim3(n_waves), dim3(64), 0, 0, buf, sink, 7u);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const int per = SHAPE == 2 ? 4 : 1;
    const double instr_per_cu = (double)ITER * per * n_waves / n_cu;
    const double ns_per_instr_cu = ms * 1e6 / instr_per_cu;       // pipeline time per wave-instruction on one CU
    printf("%-58s %s  %8.3f ms  %7.1f ns/instr/CU = %6.0f cycles @%.1f GHz\n", name, STORE ? "store" : "load ", ms, ns_per_instr_cu, ns_per_instr_cu * clk_ghz, clk_ghz);
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int wpc = argc > 1 ? atoi(argv[1]) : WAVES_PER_CU;
    const int n_waves = n_cu * wpc;
    const double clk = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %d waves (%d per CU), region %u KB per wave\n", prop.name, n_cu, n_waves, wpc, REGION / 1024);
    char *buf; uint32_t *sink;
    CHECK(hipMalloc((void **)&buf, (size_t)n_waves * REGION));
    CHECK(hipMemset(buf, 1, (size_t)n_waves * REGION));
    CHECK(hipMalloc((void **)&sink, 64));
#define BOTH(S, NAME) run<S, false>(NAME, buf, sink, n_waves, n_cu, clk); run<S, true>(NAME, buf, sink, n_waves, n_cu, clk);
    BOTH(0, "16 B/lane, one quarter of a random 64-B record per lane")
    BOTH(2, "4 x 16 B/lane, a whole random 64-B record per lane")
    BOTH(6, "16 B/lane, 4 adjacent lanes = one random 64-B record")
    BOTH(7, "16 B/lane, lanes 64 B apart (consecutive records)")
    BOTH(1, "16 B/lane contiguous (1 KB per instruction)")
    BOTH(3, "8 B/lane random")
    BOTH(5, "8 B/lane contiguous")
    BOTH(4, "4 B/lane random")
    BOTH(8, "8 B/lane, lanes 96 B apart (merge output)")
    return 0;
}
