// Dev tool: the round trip of a DEPENDENT scattered load under k_map's geometry -- every wavefront of the chip chasing through its own
// hot window -- against how far apart the windows lie (slot stride) and how large they are.  Tells a cache / TLB-reach effect of the
// slot layout from plain memory latency.
//   hipcc --offload-arch=gfx950 -O3 -o uncalled_amd/variants/ubench_latency tools/dev/ubench_latency.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// each lane: ITER dependent 8-byte loads at pseudo-random 8-byte slots of the wave's window (the loaded value feeds the next address)
__global__ __launch_bounds__(64, 4) void k_chase(char *base, size_t stride, uint32_t hot, uint32_t regions, uint32_t region_stride,
                                                 uint32_t n_slots, uint32_t hop, uint32_t iters, uint32_t *sink) {
    uint32_t slot = blockIdx.x;
    uint32_t x = blockIdx.x * 977u + threadIdx.x * 131u + 12345u;
    uint64_t acc = 0;
    const uint32_t per_region = hot / regions;       // hot bytes of a region (the first bytes of it)
    for (uint32_t it = 0; it < iters; ++it) {
        if (hop && it % hop == hop - 1) slot = (slot + gridDim.x) % n_slots;     // a parked read's slot taken up, like k_map's time slices
        x = x * 1664525u + 1013904223u + (uint32_t)acc;
        const uint32_t r = (x >> 10) % regions, o = ((x >> 3) % (per_region / 8)) * 8;
        const char *p = base + (size_t)slot * stride + (size_t)r * region_stride + o;
        acc += *(const uint64_t *)p;
    }
    if (acc == 0x123456789ull) sink[0] = 1;
}

int main(int argc, char **argv) {
    const size_t GB = 1ull << 30;
    size_t total = 40 * GB;
    char *buf; uint32_t *sink;
    CHECK(hipMalloc(&buf, total)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, total));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Cfg { const char *name; uint32_t waves; size_t stride; uint32_t hot, regions, region_stride, n_slots, hop; };
    const uint32_t K = 1024;
    std::vector<Cfg> cfgs = {
        {"dense 128K window, 4096 waves", 4096, 128 * K, 128 * K, 1, 0, 4096, 0},
        {"slot 2.2M stride, 1 region 128K", 4096, 2200 * K + 512, 128 * K, 1, 0, 4096, 0},
        {"slot 2.2M stride, 16 regions of 8K, 128K apart", 4096, 2200 * K + 512, 128 * K, 16, 128 * K, 4096, 0},
        {"same, 16384 slots hop every 256", 4096, 2200 * K + 512, 128 * K, 16, 128 * K, 16384, 256},
        {"slot 2M stride exactly, 16 regions", 4096, 2048 * K, 128 * K, 16, 128 * K, 4096, 0},
        {"dense 128K window, 1024 waves", 1024, 128 * K, 128 * K, 1, 0, 1024, 0},
        {"slot 2.2M stride 16 regions, 1024 waves", 1024, 2200 * K + 512, 128 * K, 16, 128 * K, 1024, 0},
        {"dense 32K window, 4096 waves", 4096, 32 * K, 32 * K, 1, 0, 4096, 0},
        {"slot 2.2M stride, 16 regions of 2K (32K hot)", 4096, 2200 * K + 512, 32 * K, 16, 128 * K, 4096, 0},
        {"whole-buffer random (40 GB), 4096 waves", 4096, 0, 0, 1, 0, 1, 0},
    };
    const uint32_t iters = 2000;
    for (auto &c : cfgs) {
        uint32_t hot = c.hot; size_t stride = c.stride;
        if (!hot) { hot = (uint32_t)(2 * GB - 8); }       // (one 2 GB window shared by all: random over far more than any cache / TLB)
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            k_chase<<<c.waves, 64>>>(buf, stride, hot, c.regions, c.region_stride, c.n_slots, c.hop, iters, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-52s %8.1f ms  %7.0f ns per dependent trip (%5.0f cycles at 2.4 GHz)\n", c.name, ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
        }
    }
    return 0;
}
