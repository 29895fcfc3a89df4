// k_map.hip, part 2: the bounded multi-producer/multi-consumer rings the scheduler and the node pool hand ids through.
#pragma once

namespace unc {

// ---- scheduler queues (lane 0 only).  Vyukov's bounded MPMC ring: cell.seq == pos: free for the push at pos;
// == pos + 1: holds the value of that push; a pop at pos leaves pos + cap.  At most n_slots <= cap ids exist, so a push
// never finds its cell occupied by a live value -- at worst by a pop that has not yet released it.
constexpr uint32_t SCHED_EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t ld_acq(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rel(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void sched_push(SchedQueue *q, SchedCell *cells, uint32_t mask, uint32_t v) {
    const uint32_t pos = atomicAdd(&q->tail, 1u);
    SchedCell *c = cells + (pos & mask);
    while (ld_acq(&c->seq) != pos) __builtin_amdgcn_s_sleep(1);
    c->val = v;
    st_rel(&c->seq, pos + 1u);
}

__device__ __forceinline__ uint32_t sched_pop(SchedQueue *q, SchedCell *cells, uint32_t mask) {
    for (;;) {
        const uint32_t pos = ld_acq(&q->head);
        SchedCell *c = cells + (pos & mask);
        const int32_t dif = (int32_t)(ld_acq(&c->seq) - (pos + 1u));
        if (dif < 0) return SCHED_EMPTY;
        if (dif == 0 && atomicCAS(&q->head, pos, pos + 1u) == pos) {
            const uint32_t v = c->val;
            st_rel(&c->seq, pos + mask + 1u);
            return v;
        }
    }
}

}  // namespace unc
