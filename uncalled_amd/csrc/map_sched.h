// k_map.hip, part 2: the bounded multi-producer/multi-consumer rings the scheduler and the node pool hand ids through.
#pragma once

namespace unc {

// ---- scheduler queues (lane 0 only).  Vyukov's bounded MPMC ring: cell.seq == pos: free for the push at pos;
// == pos + 1: holds the value of that push; a pop at pos leaves pos + cap.  At most n_slots <= cap ids exist, so a push
// never finds its cell occupied by a live value -- at worst by a pop that has not yet released it.
// All loads and stores of the rings are RELAXED atomics of agent scope (they go to the memory side, `sc1`): a cell's value travels in
// the same 64-bit word as its sequence number.  What a slot or a chunk POINTS at is ordered by the caller: a release fence before
// the push (`__threadfence()` where a slot is parked, tracker_release for chunks), an acquire fence after the pop (`__threadfence()`
// where a task is taken up).  An acquire per polling load was an L2 invalidate per polling load.
constexpr uint32_t SCHED_EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t ld_rlx(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t cell_ld(const SchedCell *c) {
    return __hip_atomic_load(reinterpret_cast<const uint64_t *>(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cell_st(SchedCell *c, uint32_t seq, uint32_t val) {
    __hip_atomic_store(reinterpret_cast<uint64_t *>(c), (uint64_t)seq | ((uint64_t)val << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void sched_push(SchedQueue *q, SchedCell *cells, uint32_t mask, uint32_t v) {
    const uint32_t pos = atomicAdd(&q->tail, 1u);
    SchedCell *c = cells + (pos & mask);
    while ((uint32_t)cell_ld(c) != pos) __builtin_amdgcn_s_sleep(1);
    cell_st(c, pos + 1u, v);
}

__device__ __forceinline__ uint32_t sched_pop(SchedQueue *q, SchedCell *cells, uint32_t mask) {
    for (;;) {
        const uint32_t pos = ld_rlx(&q->head);
        SchedCell *c = cells + (pos & mask);
        const uint64_t w = cell_ld(c);
        const int32_t dif = (int32_t)((uint32_t)w - (pos + 1u));
        if (dif < 0) return SCHED_EMPTY;
        if (dif == 0 && atomicCAS(&q->head, pos, pos + 1u) == pos) {
            cell_st(c, pos + mask + 1u, 0u);
            return (uint32_t)(w >> 32);
        }
        __builtin_amdgcn_s_sleep(1);      // (another pop got there first: look again)
    }
}

// ---- which pair of rings: the XCD this wavefront runs on (hardware register XCC_ID; workgroups are dealt to the XCDs round-robin, and
// the register is what says whose L2 this is), the partition that owns a slot
__device__ __forceinline__ uint32_t xcd_id() { return (uint32_t)__builtin_amdgcn_s_getreg(6164) & 0xFu; }      // hwreg(HW_REG_XCC_ID, 0, 4)
__device__ __forceinline__ uint32_t sched_part(uint32_t n_parts) { return n_parts > 1u ? xcd_id() % n_parts : 0u; }
__device__ __forceinline__ uint32_t sched_part_of_slot(uint32_t n_slots, uint32_t n_parts, uint32_t slot) { return n_parts > 1u ? slot / (n_slots / n_parts) : 0u; }

// ---- the node pool's ring (PoolQueue, unc_dev_types.h): no loop that another wavefront's progress can restart
__device__ __forceinline__ uint32_t pool_ring_pop(PoolQueue *q, SchedCell *cells, uint32_t mask) {
    const int32_t a = atomicAdd(&q->avail, -1);
    if (a <= 0) {                                    // dry: the count goes back, nothing was taken
        atomicAdd(&q->avail, 1);
        if (ld_rlx(&q->low_water) != 0u) atomicMin(&q->low_water, 0u);
        return SCHED_EMPTY;
    }
    if ((uint32_t)(a - 1) < ld_rlx(&q->low_water)) atomicMin(&q->low_water, (uint32_t)(a - 1));
    const uint32_t pos = atomicAdd(&q->head, 1u);
    SchedCell *c = cells + (pos & mask);
    uint64_t w = cell_ld(c);
    while ((uint32_t)w != pos + 1u) { __builtin_amdgcn_s_sleep(1); w = cell_ld(c); }      // (its push has its ticket and is on its way)
    cell_st(c, pos + mask + 1u, 0u);
    return (uint32_t)(w >> 32);
}
__device__ __forceinline__ void pool_ring_push(PoolQueue *q, SchedCell *cells, uint32_t mask, uint32_t v) {
    const uint32_t pos = atomicAdd(&q->tail, 1u);
    SchedCell *c = cells + (pos & mask);
    while ((uint32_t)cell_ld(c) != pos) __builtin_amdgcn_s_sleep(1);
    cell_st(c, pos + 1u, v);
    atomicAdd(&q->avail, 1);
}

}  // namespace unc
