// Wavefront (64-lane) primitives used by the kernels.  All of them are collectives: call them
// from wave-uniform control flow only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unc {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// compiler-level ordering of this wave's global/LDS traffic between cooperative phases (lanes of a
// wave share the L1 and the memory pipeline executes a wave's accesses in order)
__device__ __forceinline__ void wave_sync() {
#ifdef UNC_STRONG_SYNC
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // s_waitcnt vmcnt(0) lgkmcnt(0): stores have landed
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#endif
    __builtin_amdgcn_wave_barrier();
}

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

// segmented inclusive prefix max of u64: a lane with head != 0 starts a new segment
__device__ __forceinline__ uint64_t seg_incl_max64(uint64_t v, bool head) {
    uint64_t x = v;
    uint32_t f = head ? 1u : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t y = (uint64_t)__shfl_up((unsigned long long)x, d);
        uint32_t g = (uint32_t)__shfl_up((int)f, d);
        if (lane_id() >= d) {
            if (!f) x = x > y ? x : y;
            f |= g;
        }
    }
    return x;
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, d);
    return v;
}

// Streaming (write-once / read-once) traffic such as the path records: keep it from evicting the FM index
// out of the XCD's L2.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Value of lane (lane ^ D).  Distances 1 and 2 stay inside a quad (one DPP move per dword), 4 and 8 inside a 16-lane
// row (two bank-masked DPP row shifts per dword); none of them touches the LDS crossbar or needs a wait, unlike
// ds_bpermute, which serves the distances 16 and 32.
template <int D> __device__ __forceinline__ uint32_t xor_lane32(uint32_t v) {
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (D == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);        // banks 0,2 <- lane + 4
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);     // banks 1,3 <- lane - 4
    } else if constexpr (D == 8) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x108, 0xF, 0x3, false);        // banks 0,1 <- lane + 8
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x118, 0xF, 0xC, false);     // banks 2,3 <- lane - 8
    } else if constexpr (D == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xF, 0xF, false);   // quad_perm [3,2,1,0]
    else if constexpr (D == 7) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);    // row_half_mirror
    else if constexpr (D == 15) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);   // row_mirror
    else return (uint32_t)__shfl_xor((int)v, D);
}
template <int D> __device__ __forceinline__ uint64_t xor_lane64(uint64_t v) {
    return ((uint64_t)xor_lane32<D>((uint32_t)(v >> 32)) << 32) | xor_lane32<D>((uint32_t)v);
}
// runtime distance (uniform).  Dispatching to the DPP forms measured SLOWER than plain ds_bpermute in the sort
// networks (8.32 s against 8.04 s per 50 k reads: the five-way branch per stage and the DPP wait states cost more than
// the LDS crossbar trips they save), so the dispatch is off unless UNC_DPP_SORT is defined.
__device__ __forceinline__ uint64_t xor_lane64(uint64_t v, uint32_t d) {
#ifndef UNC_DPP_SORT
    return (uint64_t)__shfl_xor((unsigned long long)v, (int)d);
#endif
    switch (d) {
        case 1: return xor_lane64<1>(v);
        case 2: return xor_lane64<2>(v);
#ifndef UNC_DPP_QUAD_ONLY
        case 4: return xor_lane64<4>(v);
        case 8: return xor_lane64<8>(v);
#endif
        case 3: return xor_lane64<3>(v);
        case 7: return xor_lane64<7>(v);
        case 15: return xor_lane64<15>(v);
        default: return (uint64_t)__shfl_xor((unsigned long long)v, (int)d);
    }
}

// Global access as (uniform base, 32-bit byte offset): lets the compiler address with an SGPR base plus one VGPR
// (global_load ... v_off, s[base]) instead of building a 64-bit address in a VGPR pair per access.  Every per-slot
// array is far below 4 GB.
template <class T> __device__ __forceinline__ T gld(const void *base, uint32_t off) {
    return *reinterpret_cast<const T *>(static_cast<const char *>(base) + off);
}
template <class T> __device__ __forceinline__ void gst(void *base, uint32_t off, const T &v) {
    *reinterpret_cast<T *>(static_cast<char *>(base) + off) = v;
}

__device__ __forceinline__ void nt_store(uint4 *p, uint4 v) {
    u32x4 w;
    __builtin_memcpy(&w, &v, 16);
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(p));   // global_store_dwordx4 ... nt
}
__device__ __forceinline__ uint4 nt_load(const uint4 *p) {
    const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));   // global_load_dwordx4 ... nt
    uint4 v;
    __builtin_memcpy(&v, &w, 16);
    return v;
}

}  // namespace unc
