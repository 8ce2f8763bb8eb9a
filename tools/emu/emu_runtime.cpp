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
unsigned g_nthr = 0;

// block barrier state
unsigned g_live = 0, g_bar_arrived = 0, g_bar_gen = 0;
// per-wave rendezvous state
struct Wave {
    unsigned live = 0, arrived = 0, gen = 0;
    unsigned slot[64];
    bool present[64];
    unsigned snap[64];
    bool snap_present[64];
};
std::vector<Wave> g_waves;

void yield_to_scheduler() { swapcontext(&g_fibers[g_cur].ctx, &g_main); }

void trampoline()
{
    (*g_body)();
    Fiber& f = g_fibers[g_cur];
    f.done = true;
    --g_live;
    Wave& w = g_waves[g_cur / 64];
    --w.live;
    // a finished work-item no longer takes part in barriers: release waiters if it was the last one missing
    if (g_live > 0 && g_bar_arrived == g_live) { g_bar_arrived = 0; ++g_bar_gen; }
    if (w.live > 0 && w.arrived == w.live) {
        memcpy(w.snap, w.slot, sizeof(w.slot)); memcpy(w.snap_present, w.present, sizeof(w.present));
        memset(w.present, 0, sizeof(w.present)); w.arrived = 0; ++w.gen;
    }
    swapcontext(&f.ctx, &g_main);
}
}  // namespace

void emu_syncthreads()
{
    const unsigned gen = g_bar_gen;
    if (++g_bar_arrived == g_live) { g_bar_arrived = 0; ++g_bar_gen; return; }
    while (g_bar_gen == gen) yield_to_scheduler();
}

// All live lanes of the wave that reach this point exchange one 32-bit value.
unsigned emu_wave_exchange(unsigned value, unsigned* all64)
{
    Wave& w = g_waves[g_cur / 64];
    const int lane = g_cur % 64;
    w.slot[lane] = value;
    w.present[lane] = true;
    const unsigned gen = w.gen;
    if (++w.arrived == w.live) {
        memcpy(w.snap, w.slot, sizeof(w.slot)); memcpy(w.snap_present, w.present, sizeof(w.present));
        memset(w.present, 0, sizeof(w.present)); w.arrived = 0; ++w.gen;
    } else {
        while (w.gen == gen) yield_to_scheduler();
    }
    for (int i = 0; i < 64; ++i) all64[i] = w.snap_present[i] ? w.snap[i] : 0u;
    return 0;
}

void emu_wave_barrier()
{
    unsigned all[64];
    emu_wave_exchange(0u, all);
}

unsigned long long emu_ballot(bool pred)
{
    unsigned all[64];
    emu_wave_exchange(pred ? 1u : 0u, all);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) if (all[i]) m |= 1ull << i;
    return m;
}

int emu_readlane(int v, int lane)
{
    unsigned all[64];
    emu_wave_exchange((unsigned)v, all);
    return (int)all[lane & 63];
}

int emu_bpermute(int addr, int v)
{
    unsigned all[64];
    emu_wave_exchange((unsigned)v, all);
    return (int)all[(addr >> 2) & 63];
}

int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    unsigned all[64];
    emu_wave_exchange((unsigned)src, all);
    const int lane = g_cur % 64, row = lane / 16, pos = lane % 16;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (pos / 4)) & 1)) return old;
    int srcpos = -1;  // position within the row, -1 = invalid
    if (ctrl >= 0x101 && ctrl <= 0x10F) { srcpos = pos + (ctrl - 0x100); if (srcpos > 15) srcpos = -1; }       // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { srcpos = pos - (ctrl - 0x110); }                                 // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12F) { srcpos = (pos - (ctrl - 0x120) + 16) % 16; }                     // row_ror
    else if (ctrl == 0x140) { srcpos = 15 - pos; }                                                               // row_mirror
    else if (ctrl == 0x141) { srcpos = (pos & 8) | (7 - (pos & 7)); }                                            // row_half_mirror
    else if (ctrl == 0x142) { if (row == 0) return old; return (int)all[row * 16 - 1]; }                         // row_bcast:15 (lane 15 of the previous row)
    else if (ctrl == 0x143) { if (row < 2) return old; return (int)all[31]; }                                   // row_bcast:31
    else if (ctrl == 0x138) { if (lane == 0) return bound_ctrl ? 0 : old; return (int)all[lane - 1]; }          // wave_shr:1
    else if (ctrl < 0x100) { srcpos = (pos & ~3) | ((ctrl >> (2 * (pos & 3))) & 3); }                            // quad_perm
    else { fprintf(stderr, "emu: unsupported dpp ctrl 0x%x\n", ctrl); abort(); }
    if (srcpos < 0) return bound_ctrl ? 0 : old;
    return (int)all[row * 16 + srcpos];
}

void emu_launch(const char* name, dim3 grid, dim3 block, const std::function<void()>& body)
{
    const unsigned nthr = block.x * block.y * block.z;
    if (g_fibers.size() < nthr) {
        const size_t old = g_fibers.size();
        g_fibers.resize(nthr);
        for (size_t i = old; i < nthr; ++i) g_fibers[i].stack = (char*)malloc(kStack);
    }
    g_body = &body;
    g_nthr = nthr;
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = dim3(bx, by, bz);
                g_live = nthr; g_bar_arrived = 0;
                g_waves.assign((nthr + 63) / 64, Wave());
                for (unsigned t = 0; t < nthr; ++t) {
                    Fiber& f = g_fibers[t];
                    f.done = false;
                    g_waves[t / 64].live++;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &g_main;
                    makecontext(&f.ctx, trampoline, 0);
                }
                for (auto& w : g_waves) { memset(w.present, 0, sizeof(w.present)); }
                unsigned alive = nthr;
                unsigned long long spins = 0;
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
                    if (++spins > 3000000ull) { fprintf(stderr, "emu: deadlock (divergent barrier?) in %s, workgroup %u of %u\n", name, bx, grid.x); abort(); }
                }
            }
}
