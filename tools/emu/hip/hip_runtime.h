// DEVELOPMENT HARNESS ONLY (tools/emu): a tiny single-threaded SIMT emulator that lets the unmodified
// kernel sources under atracdenc_amd/csrc be compiled with g++ and stepped through on a CPU-only
// machine (this build container has no GPU). It is NOT part of the product, is never shipped in
// libat3hip.so, and no test or benchmark result is produced with it - it only shortens the
// edit/debug loop for kernel *logic* (indexing, barriers, state machines) before a gpurun call.
//
// Model: one workgroup at a time; each work-item is a ucontext fiber; __syncthreads() yields to a
// round-robin scheduler. __shared__ becomes `static` (one workgroup alive at a time).
#pragma once
#define AT3_EMU_HOST 1
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

using std::isfinite;

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 {
    float x, y;
};
struct float4 {
    float x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }


struct uint2 {
    unsigned x, y;
};
struct uint4 {
    unsigned x, y, z, w;
};
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }

extern dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
// every __shared__ object lands in one section that the scheduler poisons before each workgroup starts: a kernel that
// reads LDS it has not written sees 0xCD bytes here, not the zeros (or the previous workgroup's values) a plain static holds
#define __shared__ static __attribute__((section("emu_lds")))
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

void emu_syncthreads();
#define __syncthreads() emu_syncthreads()

static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned int __umul24(unsigned int a, unsigned int b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int __float2int_rn(float f) { return (int)lrintf(f); }
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
template <typename T, typename U> static inline T atomicAdd(T* p, U v) { T o = *p; *p = o + (T)v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }

typedef int hipError_t;
template <typename F> static inline int hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 3; return 0; }
enum { hipSuccess = 0, hipErrorUnknown = 1 };
typedef void* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

static inline const char* hipGetErrorString(hipError_t) { return "emu error"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)1; return hipSuccess; }
#define hipStreamNonBlocking 1
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); memset(*p, 0xCD, n); return *p ? hipSuccess : hipErrorUnknown; }
#define hipHostMallocDefault 0u
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; size_t maxSharedMemoryPerMultiProcessor; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 4; p->maxSharedMemoryPerMultiProcessor = 160u * 1024u; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipDeviceGetPCIBusId(char* id, int len, int) { if (len > 0) id[0] = 0; return hipErrorUnknown; }

// ---- wave-level (64 lanes) cross-lane primitives: implemented with a wave-wide rendezvous ----
unsigned emu_wave_exchange(unsigned value, unsigned* all64);   // deposit `value`, returns active mask lo; all64 = values
unsigned long long emu_ballot(bool pred);
#define __ballot(pred) emu_ballot(pred)
int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl);
int emu_bpermute(int addr, int v);
#define __builtin_amdgcn_ds_bpermute(addr, v) emu_bpermute((int)(addr), (int)(v))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_update_dpp((int)(old), (int)(src), (ctrl), (rm), (bm), (bc))
int emu_readlane(int v, int lane);
#define __builtin_amdgcn_readlane(v, lane) emu_readlane((int)(v), (lane))
static inline __attribute__((always_inline)) float __shfl(float v, int lane, int = 64) { float r; int i; memcpy(&i, &v, 4); i = emu_readlane(i, lane); memcpy(&r, &i, 4); return r; }
static inline __attribute__((always_inline)) float __shfl_xor(float v, int mask, int = 64) { return __shfl(v, (int)((threadIdx.x & 63u) ^ (unsigned)mask)); }
// v_permlane32_swap / v_permlane16_swap (gfx950): returns {a', b'}
typedef int emu_int2 __attribute__((ext_vector_type(2)));
static inline __attribute__((always_inline)) emu_int2 __builtin_amdgcn_permlane32_swap(int a, int b, bool, bool)
{
    const int ln = (int)(threadIdx.x & 63u);
    const int pa = emu_bpermute(4 * (ln ^ 32), a), pb = emu_bpermute(4 * (ln ^ 32), b);
    emu_int2 r;
    if (ln < 32) { r[0] = a; r[1] = pa; } else { r[0] = pb; r[1] = b; }
    return r;
}
static inline __attribute__((always_inline)) emu_int2 __builtin_amdgcn_permlane16_swap(int a, int b, bool, bool)
{
    const int ln = (int)(threadIdx.x & 63u);
    const int pa = emu_bpermute(4 * (ln ^ 16), a), pb = emu_bpermute(4 * (ln ^ 16), b);
    emu_int2 r;
    if ((ln & 16) == 0) { r[0] = a; r[1] = pa; } else { r[0] = pb; r[1] = b; }
    return r;
}
void emu_wave_barrier();
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return v; }
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
static inline unsigned long long __builtin_amdgcn_s_memtime() { return 0ull; }
static inline unsigned long long __builtin_amdgcn_s_memrealtime() { return 0ull; }
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned v) { const unsigned ln = threadIdx.x & 63u; return v + (unsigned)__builtin_popcount(mask & (ln >= 32 ? 0xffffffffu : ((1u << ln) - 1u))); }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned v) { const unsigned ln = threadIdx.x & 63u; return v + (ln > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (ln - 32)) - 1u)) : 0u); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }

void emu_launch(const char* name, dim3 grid, dim3 block, const std::function<void()>& body);
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu_launch(#kernel, (grid), (block), [&]() { kernel(__VA_ARGS__); })
