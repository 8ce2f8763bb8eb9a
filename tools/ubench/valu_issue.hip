// Micro-benchmark (development aid, not part of the product): what a SIMD of gfx950 issues per cycle when it PROVABLY holds four
// or eight wavefronts - the state k_alloc_pack and the QMF / MDCT kernels run in. Round 4's table (valu_lds_rates) asked the
// dispatcher for n small workgroups per CU and got 1.5 - 2.9 resident wavefronts per SIMD in its "4" rows. Here residency is by
// construction: ONE workgroup of 1024 work-items = sixteen wavefronts = four per SIMD, released together by a workgroup barrier
// (grid 256: one workgroup per CU); grid 512 with a register budget of 64 puts two such workgroups on a CU = eight per SIMD.
// Every wavefront reports the SIMD it ran on (HW_ID / XCC_ID), its loop's shader cycles (s_memtime) and its begin / end on the
// device-wide 100 MHz clock (s_memrealtime), so the table's "resident" column is counted, not asked for:
//   by id   = wavefronts that reported the same (XCC, SE, SH, CU, SIMD), averaged over the SIMDs in use
//   by time = sum of the loops' durations over (their span x SIMDs in use)
// No LDS, sixteen independent chains per wavefront, 128 instructions per loop trip.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue tools/ubench/valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP8X(B) B B B B B B B B

struct WaveRec {
    unsigned long long cycles, r0, r1;
    unsigned hw_id, xcc_id;
};

enum Kind {
    K_MUL, K_ADD, K_FMA, K_PK_MUL, K_PK_ADD, K_PK_FMA, K_MUL_ADD_DEP, K_PK_MUL_ADD_DEP, K_IADD, K_AND_OR, K_CMP_CNDMASK,
    K_VALU_SALU_2_1, K_VALU_SALU_1_1, K_SALU_ONLY, K_READLANE, K_MUL_SGPR, K_BRANCH_TAKEN, K_IF_SKELETON, K_WAITCNT
};

template <int KIND>
__global__ __launch_bounds__(1024) void k_issue(float* out, WaveRec* rec, float seed, int iters)
{
    const int tid = threadIdx.x;
    float a[16];
    f2 p[16];
    unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + (float)((tid + i) & 7) * 1e-3f;
        p[i].x = a[i];
        p[i].y = a[i] * 0.5f;
        u[i] = (unsigned)(tid * 16 + i);
    }
    const float w = seed * 0.9999f;
    f2 w2;
    w2.x = w;
    w2.y = w * 1.0001f;
    unsigned s0 = (unsigned)iters, s1 = 3u, s2 = 5u, s3 = 7u;   // scalar chains (uniform values: they live in SGPRs)
    __syncthreads();   // the sixteen wavefronts of the workgroup enter the loop together
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            REP8X(REP16(X))
#undef X
        } else if (KIND == K_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            REP8X(REP16(X))
#undef X
        } else if (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
            REP8X(REP16(X))
#undef X
        } else if (KIND == K_MUL_SGPR) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(w));
            REP8X(REP16(X))
#undef X
        } else if (KIND == K_PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(w2));
            REP8X(REP16(X))
#undef X
        } else if (KIND == K_PK_ADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(w2));
            REP8X(REP16(X))
#undef X
        } else if (KIND == K_PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(w2));
            REP8X(REP16(X))
#undef X
        } else if (KIND == K_MUL_ADD_DEP) {   // the FMA-free contract's unit of work: a product and the sum that consumes it (64 pairs)
#define X(i)                                                                          \
    {                                                                                 \
        float t;                                                                      \
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a[(i + 1) & 15]), "v"(w)); \
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t));                   \
    }
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (KIND == K_PK_MUL_ADD_DEP) {
#define X(i)                                                                               \
    {                                                                                      \
        f2 t;                                                                              \
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(p[(i + 1) & 15]), "v"(w2)); \
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(t));                    \
    }
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (KIND == K_IADD) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
            REP8X(REP16(X))
#undef X
        } else if (KIND == K_AND_OR) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
            REP8X(REP16(X))
#undef X
        } else if (KIND == K_CMP_CNDMASK) {   // a compare into VCC and the select that reads it (64 pairs): the shape of every `x ? a : b` per lane
#define X(i)                                                                                              \
    asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]) : "vcc");
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (KIND == K_VALU_SALU_2_1) {   // k_alloc_pack's mix: two vector instructions per scalar one (96 + 48 per trip; counted: 144)
#define X(i)                                                                                                                           \
    asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %3\n\tv_add_u32 %0, %0, %2" : "+v"(u[i]), "+s"(s1) : "v"(u[(i + 1) & 15]), "s"(s2) : "scc");
            REP16(X) REP16(X) REP16(X)
#undef X
        } else if (KIND == K_VALU_SALU_1_1) {   // 64 + 64
#define X(i) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(u[i]), "+s"(s1) : "v"(u[(i + 1) & 15]), "s"(s2) : "scc");
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (KIND == K_SALU_ONLY) {   // four scalar chains, 128 per trip
#define X(i) asm volatile("s_add_u32 %0, %0, %4\n\ts_xor_b32 %1, %1, %4\n\ts_add_u32 %2, %2, %4\n\ts_xor_b32 %3, %3, %4" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(iters) : "scc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_BRANCH_TAKEN) {   // a vector add and an unconditional branch over nothing (64 pairs): what a taken branch costs the wavefront
#define X(i) asm volatile("v_add_u32 %0, %0, %1\n\ts_branch 1f\n1:" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (KIND == K_IF_SKELETON) {   // the compiler's `if (lane condition) { one instruction }`: v_cmp, s_and_saveexec, s_cbranch_execz (not taken), body, s_or exec (32 x 5)
#define X(i)                                                                                                                        \
    {                                                                                                                               \
        unsigned long long sv;                                                                                                      \
        asm volatile("v_cmp_lt_u32 vcc, %2, %3\n\ts_and_saveexec_b64 %1, vcc\n\ts_cbranch_execz 1f\n\tv_add_u32 %0, %0, %2\n1:\ts_or_b64 exec, exec, %1" \
                     : "+v"(u[i]), "=&s"(sv)                                                                                        \
                     : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15])                                                                   \
                     : "vcc", "scc", "exec");                                                                                       \
    }
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_WAITCNT) {   // a vector add and an s_waitcnt that has nothing to wait for (64 pairs)
#define X(i) asm volatile("v_add_u32 %0, %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (KIND == K_READLANE) {   // v_readlane_b32 into an SGPR and a vector add that reads it (64 pairs)
#define X(i)                                                                                                                 \
    {                                                                                                                        \
        unsigned sv;                                                                                                         \
        asm volatile("v_readlane_b32 %1, %2, 3\n\ts_nop 0\n\tv_add_u32 %0, %0, %1" : "+v"(u[i]), "=&s"(sv) : "v"(u[(i + 1) & 15])); \
    }
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)(s0 + s1 + s2 + s3);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
    out[(size_t)blockIdx.x * 1024 + tid] = s;
    if ((tid & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        WaveRec& r = rec[(size_t)blockIdx.x * 16 + (tid >> 6)];
        r.cycles = t1 - t0;
        r.r0 = r0;
        r.r1 = r1;
        r.hw_id = hw;
        r.xcc_id = xcc;
    }
}

template <int KIND>
void run(const char* name, int instr_per_trip, int valu_per_trip, int wgs_per_cu, float* d_out, WaveRec* d_rec, int iters = 1024)
{
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(1024), 0, 0, d_out, d_rec, 1.0f, iters);   // warm-up: clocks, code
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(1024), 0, 0, d_out, d_rec, 1.0f, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<WaveRec> r((size_t)grid * 16);
    hipMemcpy(r.data(), d_rec, r.size() * sizeof(WaveRec), hipMemcpyDeviceToHost);
    std::map<unsigned long long, int> per_simd;
    std::map<unsigned long long, std::pair<unsigned long long, unsigned long long>> simd_span;   // first begin, last end of the SIMD's wavefronts
    std::vector<unsigned long long> starts;
    double sum_cyc = 0, sum_rt = 0;
    unsigned long long first = r[0].r0, last = r[0].r1;
    for (const WaveRec& w : r) {
        // HW_ID (gfx9): WAVE_ID [3:0], SIMD_ID [5:4], PIPE_ID [7:6], CU_ID [11:8], SH_ID [12], SE_ID [15:13]; XCC_ID [3:0] of its own register
        const unsigned long long key = ((unsigned long long)(w.xcc_id & 15u) << 16) | (w.hw_id & 0xff30u);
        per_simd[key]++;
        auto it = simd_span.find(key);
        if (it == simd_span.end()) simd_span[key] = std::make_pair(w.r0, w.r1);
        else {
            it->second.first = std::min(it->second.first, w.r0);
            it->second.second = std::max(it->second.second, w.r1);
        }
        starts.push_back(w.r0);
        sum_cyc += (double)w.cycles;
        sum_rt += (double)(w.r1 - w.r0);
        first = std::min(first, w.r0);
        last = std::max(last, w.r1);
    }
    int min_w = 1 << 30, max_w = 0;
    for (auto& kv : per_simd) {
        min_w = std::min(min_w, kv.second);
        max_w = std::max(max_w, kv.second);
    }
    const double simds = (double)per_simd.size();
    // per SIMD: the span from its first wavefront's loop begin to its last one's end (100 MHz ticks); a SIMD's throughput is its
    // wavefronts' instructions over THAT span - the device-wide span also holds the skew between the workgroups' starts
    double sum_simd_span = 0;
    for (auto& kv : simd_span) sum_simd_span += (double)(kv.second.second - kv.second.first);
    std::sort(starts.begin(), starts.end());
    const double skew_p50 = (double)(starts[starts.size() / 2] - starts[0]) / 100.0, skew_max = (double)(starts.back() - starts[0]) / 100.0;
    const double by_id = (double)r.size() / simds;
    const double by_time = sum_rt / ((double)(last - first) * simds);
    const double cyc_wave = sum_cyc / (double)r.size();
    const double mhz = sum_cyc / sum_rt * 100.0;
    const double n = (double)iters * instr_per_trip, nv = (double)iters * valu_per_trip;
    // cycles per instruction as the SIMD sees them: a wavefront's cycles per instruction / the wavefronts sharing the SIMD in time
    // cycles per instruction of a SIMD over its OWN busy span: (span in ticks x sclk / 100 MHz) / (its wavefronts x instructions each)
    const double simd_cyc = (sum_simd_span / simds) * (mhz / 100.0) / (by_id * n);
    printf("%-34s asked %d  SIMDs %4.0f  resident by id %4.2f (min %d max %d) by time %4.2f  sclk %4.0f MHz  cyc/instr: wave %6.2f  SIMD(device span) %5.2f"
           "  SIMD(own span) %5.2f  cyc/VALU/SIMD(own span) %5.2f  start skew p50 %6.1f max %6.1f us  kernel %8.1f us\n",
           name, 4 * wgs_per_cu, simds, by_id, min_w, max_w, by_time, mhz, cyc_wave / n, cyc_wave / n / by_time, simd_cyc,
           valu_per_trip ? simd_cyc * n / nv : 0.0, skew_p50, skew_max, ms * 1e3);
}

int main()
{
    float* d_out;
    WaveRec* d_rec;
    hipMalloc(&d_out, (size_t)512 * 1024 * sizeof(float));
    hipMalloc(&d_rec, (size_t)512 * 16 * sizeof(WaveRec));
    for (int w : {1, 2}) {
        run<K_MUL>("v_mul_f32", 128, 128, w, d_out, d_rec);
        run<K_ADD>("v_add_f32", 128, 128, w, d_out, d_rec);
        run<K_FMA>("v_fma_f32 (not used: contract)", 128, 128, w, d_out, d_rec);
        run<K_MUL_SGPR>("v_mul_f32 (sgpr operand)", 128, 128, w, d_out, d_rec);
        run<K_PK_MUL>("v_pk_mul_f32", 128, 128, w, d_out, d_rec);
        run<K_PK_ADD>("v_pk_add_f32", 128, 128, w, d_out, d_rec);
        run<K_PK_FMA>("v_pk_fma_f32 (not used)", 128, 128, w, d_out, d_rec);
        run<K_MUL_ADD_DEP>("v_mul_f32 -> v_add_f32 pair", 128, 128, w, d_out, d_rec);
        run<K_PK_MUL_ADD_DEP>("v_pk_mul_f32 -> v_pk_add_f32 pair", 128, 128, w, d_out, d_rec);
        run<K_IADD>("v_add_u32", 128, 128, w, d_out, d_rec);
        run<K_AND_OR>("v_and_or_b32 (VOP3)", 128, 128, w, d_out, d_rec);
        run<K_CMP_CNDMASK>("v_cmp -> v_cndmask pair", 128, 128, w, d_out, d_rec);
        run<K_VALU_SALU_2_1>("2 v_add_u32 : 1 s_add_u32", 144, 96, w, d_out, d_rec);
        run<K_VALU_SALU_1_1>("1 v_add_u32 : 1 s_add_u32", 128, 64, w, d_out, d_rec);
        run<K_SALU_ONLY>("s_add / s_xor only", 128, 0, w, d_out, d_rec);
        run<K_READLANE>("v_readlane -> v_add_u32 pair", 192, 128, w, d_out, d_rec);
        run<K_BRANCH_TAKEN>("v_add_u32 + s_branch (taken)", 128, 64, w, d_out, d_rec);
        run<K_IF_SKELETON>("if-skeleton: cmp saveexec cbr add or", 160, 64, w, d_out, d_rec);
        run<K_WAITCNT>("v_add_u32 + s_waitcnt (idle)", 128, 64, w, d_out, d_rec);
    }
    // the same with loops sixteen times as long: a fixed start-up skew between workgroups no longer weighs on the device-wide span
    for (int w : {1, 2}) {
        run<K_MUL>("LONG v_mul_f32", 128, 128, w, d_out, d_rec, 16384);
        run<K_FMA>("LONG v_fma_f32 (not used)", 128, 128, w, d_out, d_rec, 16384);
        run<K_PK_FMA>("LONG v_pk_fma_f32 (not used)", 128, 128, w, d_out, d_rec, 16384);
        run<K_MUL_ADD_DEP>("LONG v_mul_f32 -> v_add_f32 pair", 128, 128, w, d_out, d_rec, 16384);
        run<K_SALU_ONLY>("LONG s_add / s_xor only", 128, 0, w, d_out, d_rec, 16384);
    }
    return 0;
}
