// TEST INFRASTRUCTURE -- "lanesim": a tiny CPU SIMT emulator used ONLY by tests/ to run the
// product's HIP kernel sources (uncalled_amd/csrc/*.hip, unmodified, no #ifdefs in them) on the
// build container, which has no GPU.  It is NOT a fallback: the uncalled_amd package never loads
// the library built against this header, and every `-m gpu` test and bench.py run the real
// gfx950 code object.  When the kernel sources are compiled with `g++ -I tests/lanesim` this file
// shadows <hip/hip_runtime.h>:
//   * each workgroup runs as blockDim.x cooperative fibers (hand-rolled x86-64 stack switch),
//     one fiber per lane, executed round-robin on one OS thread;
//   * wave collectives (__shfl*, __ballot, ...) and __syncthreads() are rendezvous points: a lane
//     deposits its operand, yields, and resumes once every lane of the wave/block has arrived.
//     All collectives must therefore sit in wave-uniform control flow -- which the kernels
//     guarantee anyway, since HIP leaves reads from inactive lanes undefined;
//   * the HIP host API subset the library uses maps onto malloc/memcpy.
#pragma once
#include <atomic>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define LANESIM 1
#define ext_vector_type(n) vector_size(4 * (n))   /* GCC spelling of clang's 4-byte-element vectors */
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define UNC_AS_GLOBAL          /* address-space markers of the kernel sources (wave_prims.h): one flat memory here */
#define UNC_AS_CONST

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct float4 { float x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct alignas(16) ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }

namespace lanesim {
constexpr int WAVE = 64;
struct Lane {
    void *sp = nullptr;       // saved stack pointer
    char *stack = nullptr;
    dim3 tid;
    bool done = false;
    uint64_t wgen = 0, bgen = 0;   // rendezvous generations passed (wave scope / block scope)
};
struct Block {
    std::vector<Lane> lanes;
    dim3 bid, bdim, gdim;
    void *sched_sp = nullptr;
    int cur = -1;
    uint64_t wslot[2][1024];     // wave-scope operands  [generation parity][thread]
    std::function<void()> body;
};
extern thread_local Block *g_blk;
extern "C" void lanesim_switch(void **save_sp, void *load_sp);
void yield_lane();
void run_grid(dim3 grid, dim3 block, const std::function<void()> &body);
extern thread_local const void *g_kernarg;   // address of the running kernel's first argument (its argument block)
template <class T, class... R> static inline void set_kernarg(const T &first, const R &...) { g_kernarg = &first; }

// Wave-scope rendezvous: deposit `v`, wait until every live lane of this wave has deposited the
// same generation, return the table of that generation (valid until the caller's next rendezvous).
static inline const uint64_t *wave_exchange(uint64_t v) {
    Block *B = g_blk;
    int me = B->cur;
    Lane &L = B->lanes[me];
    uint64_t g = L.wgen;
    B->wslot[g & 1][me] = v;
    L.wgen = g + 1;
    int base = (me / WAVE) * WAVE, end = base + WAVE;
    if (end > (int)B->lanes.size()) end = (int)B->lanes.size();
    for (;;) {
        bool all = true;
        for (int i = base; i < end; ++i)
            if (!B->lanes[i].done && B->lanes[i].wgen <= g) { all = false; break; }
        if (all) break;
        yield_lane();
    }
    return B->wslot[g & 1];
}
static inline void block_barrier() {
    Block *B = g_blk;
    Lane &L = B->lanes[B->cur];
    uint64_t g = L.bgen;
    L.bgen = g + 1;
    for (;;) {
        bool all = true;
        for (auto &o : B->lanes)
            if (!o.done && o.bgen <= g) { all = false; break; }
        if (all) break;
        yield_lane();
    }
}
}  // namespace lanesim

#define threadIdx (lanesim::g_blk->lanes[lanesim::g_blk->cur].tid)
#define blockIdx (lanesim::g_blk->bid)
#define blockDim (lanesim::g_blk->bdim)
#define gridDim (lanesim::g_blk->gdim)
constexpr int warpSize = 64;

static inline int __lanesim_wave_base() { return (int)(threadIdx.x / 64) * 64; }
static inline int __lanesim_lane() { return (int)(threadIdx.x & 63); }

static inline void __syncthreads() { lanesim::block_barrier(); }
static inline void __builtin_amdgcn_wave_barrier() { lanesim::wave_exchange(0); }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline unsigned __builtin_amdgcn_s_getreg(int r) { return r == 6164 ? (blockIdx.x + 3u) & 7u : 0u; }      // (hardware id registers: eight XCDs dealt round-robin from XCD 3 on, CU 0)
static inline void __threadfence() {}
#define __builtin_amdgcn_fence(order, scope) std::atomic_thread_fence(std::memory_order_seq_cst)

static inline unsigned long long __ballot(int pred) {
    const uint64_t *t = lanesim::wave_exchange(pred ? 1 : 0);
    int base = __lanesim_wave_base();
    int n = (int)blockDim.x - base; if (n > 64) n = 64;
    unsigned long long m = 0;
    for (int i = 0; i < n; ++i) if (t[base + i]) m |= 1ull << i;
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) {
    int base = __lanesim_wave_base();
    int n = (int)blockDim.x - base; if (n > 64) n = 64;
    unsigned long long full = n == 64 ? ~0ull : ((1ull << n) - 1);
    return __ballot(pred) == full;
}

template <class T> static inline T __lanesim_shfl_abs(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shfl operand too wide");
    uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    const uint64_t *t = lanesim::wave_exchange(bits);
    int base = __lanesim_wave_base();
    uint64_t r = t[base + (src_lane & 63)];
    T out;
    std::memcpy(&out, &r, sizeof(T));
    return out;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = __lanesim_lane();
    int s = (l & ~(width - 1)) | (src & (width - 1));
    return __lanesim_shfl_abs(v, s);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = __lanesim_lane();
    int s = l - (int)d;
    if (s < (l & ~(width - 1))) s = l;
    return __lanesim_shfl_abs(v, s);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = __lanesim_lane();
    int s = l + (int)d;
    if (s > (l | (width - 1))) s = l;
    return __lanesim_shfl_abs(v, s);
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    return __lanesim_shfl_abs(v, __lanesim_lane() ^ mask);
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return __lanesim_shfl_abs(v, 0); }
static inline int __builtin_amdgcn_readlane(int v, int src) { return __lanesim_shfl_abs(v, src); }   // (src: wave-uniform)
// v_mov_b32_dpp as the kernels use it (row_shr:n, row_shl:n, row_bcast:15, row_bcast:31, wave_shr:1, wave_shl:1).  A lane whose
// source does not exist (or whose row / bank is masked off) keeps an undefined destination on the hardware when
// bound_ctrl is clear: the emulator hands such lanes a poison value, so that a kernel relying on it fails its parity test.
static inline int __builtin_amdgcn_mov_dpp(int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const uint64_t *t = lanesim::wave_exchange((uint64_t)(uint32_t)v);
    const int base = __lanesim_wave_base(), l = __lanesim_lane(), row = l >> 4, rl = l & 15;
    int src = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; if (rl >= n) src = l - n; }
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl - 0x100; if (rl + n < 16) src = l + n; }
    else if (ctrl == 0x142) { if (row >= 1) src = row * 16 - 1; }
    else if (ctrl == 0x143) { if (row >= 2) src = 31; }
    else if (ctrl == 0x138) { if (l >= 1) src = l - 1; }
    else if (ctrl == 0x130) { if (l < 63) src = l + 1; }
    else std::abort();
    if (!((row_mask >> row) & 1) || !((bank_mask >> (rl >> 2)) & 1)) src = -1;
    if (src < 0) return bound_ctrl ? 0 : (int)0xDEADBEEF;
    return (int)(uint32_t)t[base + src];
}

static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned acc) {
    int l = __lanesim_lane();
    return acc + (unsigned)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u)));
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned acc) {
    int l = __lanesim_lane();
    return acc + (l > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0u);
}
template <class T> static inline void __builtin_nontemporal_store(T v, T *p) { *p = v; }
template <class T> static inline T __builtin_nontemporal_load(const T *p) { return *p; }
static inline long long clock64() { return (long long)__builtin_ia32_rdtsc(); }
static inline long long wall_clock64() { return (long long)__builtin_ia32_rdtsc(); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __fma_rn(double a, double b, double c) { return std::fma(a, b, c); }
static inline float __fmaf_rn(float a, float b, float c) { return std::fmaf(a, b, c); }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }

template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (void)(*(p) = (v))
#define __HIP_MEMORY_SCOPE_WAVEFRONT 0
template <class T> static inline T __lanesim_fetch_or(T *p, T v) { T o = *p; *p = o | v; return o; }
#define __hip_atomic_fetch_or(p, v, order, scope) __lanesim_fetch_or((p), (v))
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }

// ---- host API subset -------------------------------------------------------------------------
typedef int hipError_t;
typedef void *hipStream_t;
typedef struct lanesim_event { std::chrono::steady_clock::time_point t; } *hipEvent_t;
constexpr hipError_t hipSuccess = 0;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline const char *hipGetErrorString(hipError_t) { return "lanesim"; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
static inline hipError_t hipDeviceGetPCIBusId(char *out, int cap, int dev) { std::snprintf(out, (size_t)cap, "0000:%02X:00.0", 0xC1 + dev); return 0; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = std::calloc(n ? n : 1, 1); return *p ? 0 : 2; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { std::free(p); return 0; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = std::calloc(n ? n : 1, 1); return *p ? 0 : 2; }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void **)p, n, f); }
static inline hipError_t hipHostFree(void *p) { std::free(p); return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return 0; }
static inline hipError_t hipMemset2D(void *d, size_t pitch, int v, size_t w, size_t h) {
    for (size_t i = 0; i < h; ++i) std::memset(static_cast<char *>(d) + i * pitch, v, w);
    return 0;
}
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t i = 0; i < h; ++i) std::memcpy(static_cast<char *>(d) + i * dp, static_cast<const char *>(s) + i * sp, w);
    return 0;
}
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new lanesim_event(); return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return 0;
}
struct hipDeviceProp_t { int multiProcessorCount; char name[256]; char gcnArchName[256]; size_t totalGlobalMem; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    std::memset(p, 0, sizeof *p);
    p->multiProcessorCount = 4;
    std::strcpy(p->name, "lanesim");
    std::strcpy(p->gcnArchName, "lanesim");
    p->totalGlobalMem = 1ull << 34;
    return 0;
}
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = 1ull << 33; *t = 1ull << 34; return 0; }
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1 };
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 100000; return 0; }
struct hipFuncAttributes { int numRegs; size_t localSizeBytes, sharedSizeBytes; int maxThreadsPerBlock; };
static inline hipError_t hipFuncGetAttributes(hipFuncAttributes *a, const void *) { a->numRegs = 0; a->localSizeBytes = a->sharedSizeBytes = 0; a->maxThreadsPerBlock = 64; return 0; }
struct hipPointerAttribute_t { int type; };
constexpr int hipMemoryTypeHost = 0, hipMemoryTypeDevice = 1;
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeDevice; return 0; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    (lanesim::set_kernarg(__VA_ARGS__), lanesim::run_grid(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); }))
static inline const void *__builtin_amdgcn_kernarg_segment_ptr() { return lanesim::g_kernarg; }
