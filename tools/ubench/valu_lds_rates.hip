// Micro-benchmark (development aid, not part of the product): issue cost of the instructions the ATRAC3 kernels are
// built from, in shader cycles per wave64 instruction per SIMD, at 1, 2, 4, 6 and 8 resident waves per SIMD,
// with the shader clock observed under each load (s_memtime cycles against the 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o valu_lds_rates tools/ubench/valu_lds_rates.hip && ./valu_lds_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 2048;
constexpr int kUnroll = 16;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, long long* cyc, float seed, int iters)
{
    // Dynamic LDS sized by the host so that EXACTLY n workgroups fit a CU (160 KB / n each): the dispatcher has to give every
    // CU its n - with a small footprint it stacks several workgroups on some CUs and leaves others empty, and the waves per
    // SIMD are not what the row says. (Round 2's version held a fixed 64 KB, so its "4 waves per SIMD" rows really ran as two
    // rounds of two.) Only the first 4 KB are touched.
    extern __shared__ __attribute__((aligned(16))) float s_buf[];
    const int tid = threadIdx.x;
    float a[16];
    f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + (float)(tid + i);
        p[i].x = a[i];
        p[i].y = a[i] * 0.5f;
    }
    for (int i = tid; i < 1024; i += 256) s_buf[i] = seed;
    f4 acc4 = {0.f, 0.f, 0.f, 0.f};
    f2 acc2 = {0.f, 0.f};
    float acc1 = 0.f;
    const float w = seed * 0.999f;
    f2 w2;
    w2.x = w;
    w2.y = w * 1.0001f;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();   // constant 100 MHz reference
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            REP16(X)
#undef X
        } else if (KIND == 1) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            REP16(X)
#undef X
        } else if (KIND == 2) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
            REP16(X)
#undef X
        } else if (KIND == 3) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(w2));
            REP16(X)
#undef X
        } else if (KIND == 4) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(w2));
            REP16(X)
#undef X
        } else if (KIND == 5) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(w2));
            REP16(X)
#undef X
        } else if (KIND == 6) {   // mul with a scalar (SGPR) operand, as the QMF taps are
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "s"(w2));
            REP16(X)
#undef X
        } else if (KIND == 7) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == 8) {
#define X(i) asm volatile("v_mul_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(w));
            REP16(X)
#undef X
        } else if (KIND == 9) {   // 16-byte LDS reads, conflict free
            const float4* q = reinterpret_cast<const float4*>(s_buf) + tid;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                f4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)q));
                acc4.x += v.x;
            }
        } else if (KIND == 10) {
            const float2* q = reinterpret_cast<const float2*>(s_buf) + tid;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                f2 v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)q));
                acc2.x += v.x;
            }
        } else if (KIND == 11) {
            const float* q = s_buf + tid;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v;
                asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)q));
                acc1 += v;
            }
        } else if (KIND == 12) {
            float2* q = reinterpret_cast<float2*>(s_buf) + tid;
            f2 v = {a[0], a[1]};
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)q), "v"(v) : "memory");
        } else if (KIND == 13) {
            float4* q = reinterpret_cast<float4*>(s_buf) + tid;
            f4 v = {a[0], a[1], a[2], a[3]};
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)q), "v"(v) : "memory");
        } else if (KIND == 14) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int v = __builtin_amdgcn_ds_bpermute(4 * ((tid + 1) & 63), __float_as_int(a[i]));
                a[i] = __int_as_float(v);
            }
        } else if (KIND == 15) {   // QMF inner step: packed multiply by a scalar tap pair + packed add (no FMA)
#define X(i)                                                                      \
    {                                                                             \
        f2 t;                                                                     \
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(p[(i + 1) & 15]), "s"(w2)); \
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(t));          \
    }
            REP16(X)
#undef X
        } else if (KIND == 16) {   // the same with plain ops (two lanes' worth per step is twice as many instructions)
#define X(i)                                                                   \
    {                                                                          \
        float t;                                                               \
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a[(i + 1) & 15]), "s"(w)); \
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t));          \
    }
            REP16(X)
#undef X
        } else if (KIND == 17) {
#define X(i) asm volatile("v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(w) : "vcc");
            REP16(X)
#undef X
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    const long long r1 = (long long)__builtin_amdgcn_s_memrealtime();
    float s = acc4.x + acc2.x + acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) {
        cyc[(blockIdx.x * 4 + (tid >> 6)) * 4] = t1 - t0;
        cyc[(blockIdx.x * 4 + (tid >> 6)) * 4 + 1] = r1 - r0;
        cyc[(blockIdx.x * 4 + (tid >> 6)) * 4 + 2] = r0;   // when the wavefront's loop began and ended (100 MHz reference, one clock for the whole device)
        cyc[(blockIdx.x * 4 + (tid >> 6)) * 4 + 3] = r1;
    }
}

template <int KIND>
void run(const char* name, int per_iter_instr, int wgs_per_cu, float* d_out, long long* d_cyc, int iters = kIters)
{
    const int grid = 256 * wgs_per_cu;   // 4 waves per workgroup = one per SIMD
    size_t lds = (size_t)(160 * 1024 / wgs_per_cu) & ~(size_t)1023;
    if (lds * (wgs_per_cu + 1) <= 160 * 1024) lds += 1024;
    if (lds > 160 * 1024) lds = 160 * 1024;
    if (wgs_per_cu == 1) lds = 96 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rate<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), lds, 0, d_out, d_cyc, 1.0f, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), lds, 0, d_out, d_cyc, 1.0f, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(grid * 4 * 4);
    hipMemcpy(c.data(), d_cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0, avg_rt = 0;
    long long first = c[2], last = c[3];
    for (size_t i = 0; i < c.size(); i += 4) {
        avg += (double)c[i];
        avg_rt += (double)c[i + 1];
        first = c[i + 2] < first ? c[i + 2] : first;
        last = c[i + 3] > last ? c[i + 3] : last;
    }
    // how many wavefronts REALLY shared a SIMD on average: the wavefronts' loop times added up, over the span from the first
    // loop's begin to the last one's end, per SIMD (the dispatcher does not always place what the grid asks for side by side)
    const double resident = avg_rt * 0.0 + (avg_rt > 0 ? (avg_rt) : 0);   // (sum below)
    (void)resident;
    const double sum_rt = avg_rt;
    avg /= c.size() / 4;
    avg_rt /= c.size() / 4;
    const double true_waves = sum_rt / ((double)(last - first) * 1024.0);
    const double n = (double)iters * per_iter_instr;
    // s_memtime ticks = shader cycles, s_memrealtime = 100 MHz: their ratio is the shader clock UNDER THIS LOAD; cycles per
    // wave-instruction per SIMD = the wave's cycles per instruction / the waves that share the SIMD
    const double mhz = avg / avg_rt * 100.0;
    printf("%-30s waves/SIMD asked %d resident %4.2f  sclk=%5.0f MHz  cyc/instr(wave)=%7.3f  cyc/instr/SIMD=%6.3f (= %5.3f ns)  loops span %8.2f us, kernel %8.2f us\n", name,
           wgs_per_cu, true_waves, mhz, avg / n, avg / n / true_waves, avg / n / true_waves / mhz * 1e3, (double)(last - first) / 100.0, ms * 1e3);
}

int main()
{
    float* d_out;
    long long* d_cyc;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&d_cyc, 256 * 8 * 4 * 4 * sizeof(long long));
    for (int w : {1, 2, 3, 4, 6, 8}) {
        run<0>("v_mul_f32", 16, w, d_out, d_cyc);
        run<1>("v_add_f32", 16, w, d_out, d_cyc);
        run<2>("v_fma_f32", 16, w, d_out, d_cyc);
        run<3>("v_pk_mul_f32", 16, w, d_out, d_cyc);
        run<4>("v_pk_add_f32", 16, w, d_out, d_cyc);
        run<5>("v_pk_fma_f32", 16, w, d_out, d_cyc);
        run<6>("v_pk_mul_f32 (sgpr operand)", 16, w, d_out, d_cyc);
        run<7>("v_mov_b32_dpp quad_perm", 16, w, d_out, d_cyc);
        run<8>("v_mul_f32_dpp row_shr:1", 16, w, d_out, d_cyc);
        run<17>("v_cndmask_b32_dpp quad_perm", 16, w, d_out, d_cyc);
        run<9>("ds_read_b128", 16, w, d_out, d_cyc);
        run<10>("ds_read_b64", 16, w, d_out, d_cyc);
        run<11>("ds_read_b32", 16, w, d_out, d_cyc);
        run<12>("ds_write_b64", 16, w, d_out, d_cyc);
        run<13>("ds_write_b128", 16, w, d_out, d_cyc);
        run<14>("ds_bpermute_b32", 16, w, d_out, d_cyc);
        run<15>("pk_mul(sgpr)+pk_add pair", 32, w, d_out, d_cyc);
        run<16>("mul(sgpr)+add pair (plain)", 32, w, d_out, d_cyc);
    }
    // sustained load (tens of milliseconds per launch): the clock the part settles at under pure vector work
    for (int w : {4, 8}) {
        run<0>("SUSTAINED v_mul_f32", 16, w, d_out, d_cyc, kIters * 64);
        run<3>("SUSTAINED v_pk_mul_f32", 16, w, d_out, d_cyc, kIters * 64);
        run<15>("SUSTAINED pk_mul+pk_add pair", 32, w, d_out, d_cyc, kIters * 64);
        run<10>("SUSTAINED ds_read_b64", 16, w, d_out, d_cyc, kIters * 16);
    }
    return 0;
}
