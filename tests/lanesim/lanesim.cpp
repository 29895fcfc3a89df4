// TEST INFRASTRUCTURE -- scheduler of the lanesim CPU SIMT emulator (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>

namespace lanesim {

thread_local Block *g_blk = nullptr;
thread_local const void *g_kernarg = nullptr;

// void lanesim_switch(void **save_sp, void *load_sp): save callee-saved regs on the current
// stack, publish the stack pointer, adopt the other stack, restore its registers, return into it.
__asm__(
    ".text\n"
    ".globl lanesim_switch\n"
    ".type lanesim_switch,@function\n"
    "lanesim_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size lanesim_switch,.-lanesim_switch\n");

static void lane_entry() {
    Block *B = g_blk;
    B->body();
    B->lanes[B->cur].done = true;
    // return to the scheduler for good
    void *dummy;
    lanesim_switch(&dummy, B->sched_sp);
    std::abort();
}

void yield_lane() {
    Block *B = g_blk;
    Lane &L = B->lanes[B->cur];
    lanesim_switch(&L.sp, B->sched_sp);
}

constexpr size_t STACK_BYTES = 512 * 1024;

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
    unsigned nthreads = block.x * block.y * block.z;
    assert(block.y == 1 && block.z == 1 && nthreads <= 1024);
    Block B;
    B.bdim = block;
    B.gdim = grid;
    B.body = body;
    B.lanes.resize(nthreads);
    for (auto &L : B.lanes) L.stack = (char *)std::aligned_alloc(64, STACK_BYTES);
    Block *saved = g_blk;
    g_blk = &B;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                B.bid = dim3(bx, by, bz);
                for (unsigned t = 0; t < nthreads; ++t) {
                    Lane &L = B.lanes[t];
                    L.tid = dim3(t, 0, 0);
                    L.done = false;
                    L.wgen = L.bgen = 0;
                    // craft an initial frame: six zeroed callee-saved registers, then the entry address
                    uintptr_t top = ((uintptr_t)(L.stack + STACK_BYTES)) & ~(uintptr_t)63;
                    void **sp = (void **)top;
                    *--sp = nullptr;               // fake return address for alignment (entry never returns)
                    *--sp = (void *)&lane_entry;   // `ret` target
                    for (int i = 0; i < 6; ++i) *--sp = nullptr;
                    L.sp = (void *)sp;
                }
                unsigned live = nthreads;
                while (live) {
                    live = 0;
                    for (unsigned t = 0; t < nthreads; ++t) {
                        Lane &L = B.lanes[t];
                        if (L.done) continue;
                        B.cur = (int)t;
                        lanesim_switch(&B.sched_sp, L.sp);
                        if (!L.done) ++live;
                    }
                }
            }
    g_blk = saved;
    for (auto &L : B.lanes) std::free(L.stack);
}

}  // namespace lanesim
