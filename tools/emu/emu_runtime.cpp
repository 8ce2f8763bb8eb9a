// DEVELOPMENT HARNESS ONLY - see hip/hip_runtime.h in this directory.
#include "hip/hip_runtime.h"

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
};
constexpr size_t kStack = 256 * 1024;
std::vector<Fiber> g_fibers;
ucontext_t g_main;
const std::function<void()>* g_body = nullptr;
int g_cur = -1;

void trampoline()
{
    (*g_body)();
    g_fibers[g_cur].done = true;
    swapcontext(&g_fibers[g_cur].ctx, &g_main);
}
}  // namespace

void emu_syncthreads() { swapcontext(&g_fibers[g_cur].ctx, &g_main); }

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
    const unsigned nthr = block.x * block.y * block.z;
    if (g_fibers.size() < nthr) {
        const size_t old = g_fibers.size();
        g_fibers.resize(nthr);
        for (size_t i = old; i < nthr; ++i) g_fibers[i].stack = (char*)malloc(kStack);
    }
    g_body = &body;
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = dim3(bx, by, bz);
                for (unsigned t = 0; t < nthr; ++t) {
                    Fiber& f = g_fibers[t];
                    f.done = false;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &g_main;
                    makecontext(&f.ctx, trampoline, 0);
                }
                unsigned alive = nthr;
                while (alive) {
                    alive = 0;
                    for (unsigned t = 0; t < nthr; ++t) {
                        Fiber& f = g_fibers[t];
                        if (f.done) continue;
                        g_cur = (int)t;
                        threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                        swapcontext(&g_main, &f.ctx);
                        if (!f.done) ++alive;
                    }
                }
            }
}
