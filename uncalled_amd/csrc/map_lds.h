// k_map.hip, part 1: the LDS of a wavefront and the kernel's argument block.  Included by k_map.hip only (the __shared__
// arrays are DEFINED here: one translation unit).
#pragma once

namespace unc {

// ---- LDS of a wavefront (a workgroup is one wavefront): file scope, so that the phases of an event, which are separate
// functions, all see it as LDS ----
constexpr int CAND_MAX = 4 * WAVE;    // candidate (parent, base) pairs per pass
constexpr int CHILD_MAX = 5 * WAVE;   // children per pass
__shared__ __attribute__((aligned(16))) float s_probs[NKMER];      // match log-probs of the current event
__shared__ uint32_t s_flags[NKMER / 32];                            // sources_added_
// one carved buffer for the per-pass staging of phase E, reused as the merge tile of the sort and the source list of phase F:
// with the probs table a wavefront stays under 10 KB of LDS, so that 16 fit a CU
// (phase E, 32-bit rows: FM results 2 KB | parents' rows 512 B | history 512 | last / sub / moves / meta 4 x 256 | child
// descriptors 640 | the children's run positions 640; the candidate list shares the last two, which are written after it is dead)
constexpr uint32_t S_E_WORDS = (CAND_MAX * 8 + 2 * WAVE * 4 + WAVE * 8 + 4 * WAVE * 4 + 2 * CHILD_MAX * 2) / 8;
__shared__ __attribute__((aligned(16))) uint64_t s_e[S_E_WORDS];

struct MapArgs {
    DevIndex ix;
    DevScratch sc;
    DevReads rd;
    unc_params_t P;
    DevResult *results;
    uint32_t *next_read;    // work-queue head
    uint32_t max_steps;     // map_next calls per launch (0xFFFFFFFF = run to completion)
    uint32_t resume;        // 1: continue the read saved in SlotState (trace / chunked mode)
    const uint32_t *read_list;  // batch mode: the queue hands out read_list[t] instead of t (re-runs of selected reads)
    const uint32_t *slot_map;   // resume mode: block b works on scratch slot slot_map[b] (null: slot = b), descriptor b
    unsigned long long *wave_ticks;   // optional: sum over waves of (exit - start) in wall_clock64 ticks (queue-tail probe)
    const uint32_t *flags_in;   // batch mode, optional: sources_added_ a read STARTS with, [read][NKMER / 32] (null: clear)
    uint32_t *flags_out;        // batch mode, optional: sources_added_ as the read leaves it, same shape
    DevSched sched;             // batch mode with sched.ctl != null: slots are handed out per task, max_steps = slice length
    DevPool pool;               // nodes of the seed-cluster grids
};

}  // namespace unc
