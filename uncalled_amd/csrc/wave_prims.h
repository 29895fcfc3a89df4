// Wavefront (64-lane) primitives used by the kernels.  All of them are collectives: call them
// from wave-uniform control flow only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Invariants that only the CPU emulator of the tests checks (tests/lanesim defines LANESIM): an index about to be used for a store
// lies inside the buffer it addresses.  Compiled out of the gfx950 build.
#ifdef LANESIM
#include <cassert>
#define UNC_SIM_CHECK(c) assert(c)
#else
#define UNC_SIM_CHECK(c) ((void)0)
#endif

namespace unc {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// compiler-level ordering of this wave's global/LDS traffic between cooperative phases (lanes of a
// wave share the L1 and the memory pipeline executes a wave's accesses in order)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Drop this CU's vector L1: data another CU of the SAME XCD has written since (through the L2 they share) is read afresh.  The
// agent-scope acquire does this too, and invalidates the XCD's whole L2 on top.
#ifdef LANESIM
__device__ __forceinline__ void l1_invalidate() {}
#else
__device__ __forceinline__ void l1_invalidate() { asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc0" ::: "memory"); }
#endif

__device__ __forceinline__ uint32_t bcast32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src); }
__device__ __forceinline__ uint64_t bcast64(uint64_t v, int src) { return (uint64_t)__shfl((unsigned long long)v, src); }
__device__ __forceinline__ float bcastf(float v, int src) { return __shfl(v, src); }
__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    uint32_t lo = uniform32((uint32_t)v), hi = uniform32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// number of set bits of `mask` below this lane (v_mbcnt_lo/hi: two instructions, no cross-lane traffic)
__device__ __forceinline__ uint32_t prefix_popc(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// exclusive prefix sum of per-lane counts < 2^BITS: one ballot + mbcnt per bit plane
template <int BITS>
__device__ __forceinline__ uint32_t excl_sum_bits(uint32_t v, uint32_t *total) {
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const uint64_t m = __ballot((v >> b) & 1u);
        pre += prefix_popc(m) << b;
        tot += (uint32_t)__popcll(m) << b;
    }
    *total = tot;
    return pre;
}

// exclusive prefix sum of small per-lane counts; *total = sum over the wave
__device__ __forceinline__ uint32_t excl_sum32(uint32_t v, uint32_t *total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = (uint32_t)__shfl_up((int)x, d);
        if (lane_id() >= d) x += y;
    }
    *total = bcast32(x, 63);
    return x - v;
}

// exclusive prefix max (identity 0); *total = max over the wave
__device__ __forceinline__ uint32_t excl_max32(uint32_t v, uint32_t *total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = (uint32_t)__shfl_up((int)x, d);
        if (lane_id() >= d) x = x > y ? x : y;
    }
    *total = bcast32(x, 63);
    uint32_t e = (uint32_t)__shfl_up((int)x, 1);
    return lane_id() == 0 ? 0u : e;
}

// v_mov_b32_dpp: the value of another lane without a trip through the LDS crossbar.  CTRL: 0x110 + n row_shr:n (lane - n inside
// its row of 16), 0x142 row_bcast:15 (lane 15 of the previous row), 0x143 row_bcast:31 (lane 31, for lanes 32..63).  A lane
// without a source gets an undefined value: callers guard or overwrite it.  (wave_shr:1 / wave_shl:1, 0x138 / 0x130, exist on
// gfx950 and give right answers, but the walk's neighbour look-ups got slower with them than with ds_bpermute: +3.5 % on k_map.)
template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ uint64_t dpp_mov64(uint64_t v) {
    return ((uint64_t)dpp_mov32<CTRL>((uint32_t)(v >> 32)) << 32) | dpp_mov32<CTRL>((uint32_t)v);
}

// segmented inclusive prefix max of u64: a lane with head != 0 starts a new segment.  Rows of 16 lanes are scanned with
// row_shr 1 / 2 / 4 / 8, then rows 1 and 3 take in lane 15 / 47 and the upper half takes in lane 31: six steps of three
// DPP moves, no LDS traffic (round 2, 50 k E. coli reads, same box: k_map 6304 -> 6263 ms against the ds_bpermute form)
__device__ __forceinline__ uint64_t seg_incl_max64(uint64_t v, bool head) {
    uint64_t x = v;
    uint32_t f = head ? 1u : 0u;
    const uint32_t l = (uint32_t)lane_id(), rl = l & 15u;
#define UNC_SEG_STEP(CTRL, GUARD)                                   \
    {                                                               \
        const uint64_t y = dpp_mov64<CTRL>(x);                      \
        const uint32_t g = dpp_mov32<CTRL>(f);                      \
        if (GUARD) {                                                \
            if (!f) x = x > y ? x : y;                              \
            f |= g;                                                 \
        }                                                           \
    }
    UNC_SEG_STEP(0x111, rl >= 1u)
    UNC_SEG_STEP(0x112, rl >= 2u)
    UNC_SEG_STEP(0x114, rl >= 4u)
    UNC_SEG_STEP(0x118, rl >= 8u)
    UNC_SEG_STEP(0x142, (l & 31u) >= 16u)
    UNC_SEG_STEP(0x143, l >= 32u)
#undef UNC_SEG_STEP
    return x;
}


// the same scan for 32-bit values: two DPP moves per step
__device__ __forceinline__ uint32_t seg_incl_max32(uint32_t v, bool head) {
    uint32_t x = v;
    uint32_t f = head ? 1u : 0u;
    const uint32_t l = (uint32_t)lane_id(), rl = l & 15u;
#define UNC_SEG_STEP32(CTRL, GUARD)                                 \
    {                                                               \
        const uint32_t y = dpp_mov32<CTRL>(x);                      \
        const uint32_t g = dpp_mov32<CTRL>(f);                      \
        if (GUARD) {                                                \
            if (!f) x = x > y ? x : y;                              \
            f |= g;                                                 \
        }                                                           \
    }
    UNC_SEG_STEP32(0x111, rl >= 1u)
    UNC_SEG_STEP32(0x112, rl >= 2u)
    UNC_SEG_STEP32(0x114, rl >= 4u)
    UNC_SEG_STEP32(0x118, rl >= 8u)
    UNC_SEG_STEP32(0x142, (l & 31u) >= 16u)
    UNC_SEG_STEP32(0x143, l >= 32u)
#undef UNC_SEG_STEP32
    return x;
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, d);
    return v;
}

// Value of lane (lane ^ d), d uniform.  (DPP forms of the short distances were tried in the sort networks and measured
// slower than ds_bpermute there: the per-stage dispatch on d and the DPP wait states cost more than the crossbar trips.)
__device__ __forceinline__ uint64_t xor_lane64(uint64_t v, uint32_t d) {
    return (uint64_t)__shfl_xor((unsigned long long)v, (int)d);
}

// ---- where the wavefront waits for memory.  The compiler places s_waitcnt at the first USE of a loaded register, and when the number
// of memory operations issued since the load is not known at compile time (a loop, a branch around another load or store) the wait
// it emits is vmcnt(0): for EVERYTHING outstanding, the stores just issued included (a store's acknowledgement takes as long as a
// load).  Read off the ISA in round 5: the "prefetched" parent records of phase E, the info words of the walk and the staged merge
// tiles were each waited for with vmcnt(0) at a point where fresh loads or stores were in flight, so that every pass paid three or
// four memory round trips in a row instead of one.  mem_retire() is an empty asm statement that USES the value: put where the
// memory counter is zero anyway (right after a wait that cannot be avoided), it tells the compiler's bookkeeping that the value has
// arrived, and the later use costs no wait.  No instruction is emitted.  (The emulator needs none of it.)
// (16-byte values as ONE register tuple: an asm operand per component makes the compiler copy the components out of the tuple right
// behind the load, which is a use, which is a wait)
#ifdef LANESIM
typedef uint4 u32x4_t;
typedef float4 f32x4_t;
template <class T> __device__ __forceinline__ void mem_retire(T &) {}
#else
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mem_retire(u32x4_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void mem_retire(f32x4_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void mem_retire(uint32_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void mem_retire(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void mem_retire(uint64_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void mem_retire(uint4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void mem_retire(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void mem_retire(ulonglong2 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
#endif

// Address spaces are spelled out wherever a pointer travels through a function call: a generic pointer that the compiler
// cannot trace to its origin costs FLAT instructions (no scalar base, both memory counters), a kernel argument read through
// a generic pointer costs vector loads.  (The CPU emulator of tests/ defines both markers empty.)
#ifndef UNC_AS_GLOBAL
#define UNC_AS_GLOBAL __attribute__((address_space(1)))
#define UNC_AS_CONST __attribute__((address_space(4)))
#endif
typedef UNC_AS_GLOBAL char *gptr_t;            // global memory (a read's scratch slot, the node pool)
typedef const UNC_AS_GLOBAL char *cgptr_t;

// Global access as (uniform base, 32-bit byte offset): lets the compiler address with an SGPR base plus one VGPR
// (global_load ... v_off, s[base]) instead of building a 64-bit address in a VGPR pair per access.  Every per-slot
// array is far below 4 GB.
// (structs are copied with the memcpy builtin, which takes pointers of any address space and becomes one load / store
// of the struct's size and alignment; a class-type lvalue in a named address space has no copy constructor to bind to)
template <class T> __device__ __forceinline__ T g_load(const UNC_AS_GLOBAL T *p) {
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}
template <class T> __device__ __forceinline__ void g_store(UNC_AS_GLOBAL T *p, const T &v) { __builtin_memcpy(p, &v, sizeof(T)); }
template <class T> __device__ __forceinline__ T gld(cgptr_t base, uint32_t off) { return g_load(reinterpret_cast<const UNC_AS_GLOBAL T *>(base + off)); }
template <class T> __device__ __forceinline__ void gst(gptr_t base, uint32_t off, const T &v) { g_store(reinterpret_cast<UNC_AS_GLOBAL T *>(base + off), v); }
// a pointer handed to an out-of-line function arrives in vector registers: back to a scalar pair
template <class P> __device__ __forceinline__ P uniform_ptr(P p) {
    return (P)uniform64((uint64_t)p);
}
// a struct of the kernel's argument block (constant address space) by value: only the fields that are used get loaded
template <class T> __device__ __forceinline__ T ka_get(const UNC_AS_CONST T *p) {
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}

}  // namespace unc
