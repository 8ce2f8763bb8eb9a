// DEVELOPMENT HARNESS ONLY - see hip/hip_runtime.h in this directory.
#include "hip/hip_runtime.h"
#include <dlfcn.h>
// k_alloc_pack's harness-only counters (at3_k_alloc.hpp: AT3_STAT)
extern "C" { unsigned long long g_alloc_stats[16] = {0}; }

dim3 threadIdx, blockIdx, blockDim, gridDim;

extern "C" char __start_emu_lds[] __attribute__((weak)), __stop_emu_lds[] __attribute__((weak));

namespace {
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    const char* wait = "";   // what the work-item waits in, and where it was called from (deadlock report)
    void* wait_pc = nullptr;
    void* hist[16] = {};   // the last call sites, newest at hist_n % 16
    unsigned long long hist_n = 0;
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
    void* site[64];
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

#define EMU_WAIT(what) do { Fiber& f_ = g_fibers[g_cur]; f_.wait = what; f_.wait_pc = __builtin_return_address(0); f_.hist[++f_.hist_n % 16] = f_.wait_pc; } while (0)

__attribute__((noinline)) void emu_syncthreads()
{
    EMU_WAIT("__syncthreads");
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
    w.site[lane] = g_fibers[g_cur].wait_pc;
    const unsigned gen = w.gen;
    if (++w.arrived == w.live) {
        // every lane of a rendezvous must come from the same call: lanes that meet from two different ones have
        // taken different paths through code this harness can only run convergently
        // (EMU_STRICT=1, for -O0 builds: an optimiser may duplicate one call into two places)
        static const bool strict = getenv("EMU_STRICT") != nullptr;
        for (int i = 0; strict && i < 64; ++i)
            if (w.present[i] && w.site[i] != w.site[lane]) {
                Dl_info di;
                const bool ok = dladdr(w.site[lane], &di) != 0;
                char* base = ok ? (char*)di.dli_fbase : (char*)0;
                fprintf(stderr, "emu: divergent rendezvous: lane %d called from +0x%zx, lane %d from +0x%zx\n", lane, (size_t)((char*)w.site[lane] - base), i, (size_t)((char*)w.site[i] - base));
                abort();
            }
        memcpy(w.snap, w.slot, sizeof(w.slot)); memcpy(w.snap_present, w.present, sizeof(w.present));
        memset(w.present, 0, sizeof(w.present)); w.arrived = 0; ++w.gen;
    } else {
        while (w.gen == gen) yield_to_scheduler();
    }
    for (int i = 0; i < 64; ++i) all64[i] = w.snap_present[i] ? w.snap[i] : 0u;
    return 0;
}

__attribute__((noinline)) void emu_wave_barrier()
{
    EMU_WAIT("wave_sync");
    unsigned all[64];
    emu_wave_exchange(0u, all);
}

__attribute__((noinline)) unsigned long long emu_ballot(bool pred)
{
    EMU_WAIT("ballot");
    unsigned all[64];
    emu_wave_exchange(pred ? 1u : 0u, all);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) if (all[i]) m |= 1ull << i;
    return m;
}

__attribute__((noinline)) int emu_readlane(int v, int lane)
{
    EMU_WAIT("readlane");
    unsigned all[64];
    emu_wave_exchange((unsigned)v, all);
    return (int)all[lane & 63];
}

__attribute__((noinline)) int emu_bpermute(int addr, int v)
{
    EMU_WAIT("bpermute");
    unsigned all[64];
    emu_wave_exchange((unsigned)v, all);
    return (int)all[(addr >> 2) & 63];
}

__attribute__((noinline)) int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    EMU_WAIT("dpp");
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
                if (&__start_emu_lds[0] != nullptr && &__stop_emu_lds[0] > &__start_emu_lds[0]) memset(__start_emu_lds, 0xCD, (size_t)(__stop_emu_lds - __start_emu_lds));
                g_live = nthr; g_bar_arrived = 0;
                g_waves.assign((nthr + 63) / 64, Wave());
                for (unsigned t = 0; t < nthr; ++t) {
                    Fiber& f = g_fibers[t];
                    f.done = false; f.hist_n = 0;
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
                    if (++spins > 300000ull) { fprintf(stderr, "emu: deadlock (divergent barrier?) in %s, workgroup %u of %u\n", name, bx, grid.x);
                        for (unsigned t = 0; t < nthr; ++t)
                            if (!g_fibers[t].done) {
                                Dl_info di;   // offset in the shared object: addr2line -e <lib> <offset>
                                const bool ok = dladdr(g_fibers[t].wait_pc, &di) != 0;
                                fprintf(stderr, "  work-item %u waits in %s called from +0x%zx; %llu rendezvous so far, the last ones from", t, g_fibers[t].wait, ok ? (size_t)((char*)g_fibers[t].wait_pc - (char*)di.dli_fbase) : (size_t)g_fibers[t].wait_pc, g_fibers[t].hist_n);
                                for (unsigned h = 0; h < 16 && h < g_fibers[t].hist_n; ++h)
                                    fprintf(stderr, " +0x%zx", (size_t)((char*)g_fibers[t].hist[(g_fibers[t].hist_n - h) % 16] - (ok ? (char*)di.dli_fbase : (char*)0)));
                                fprintf(stderr, "\n");
                            }
                        abort();
                    }
                }
            }
}
