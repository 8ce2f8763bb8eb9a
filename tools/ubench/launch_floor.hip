// Micro-benchmark (development aid): duration of a kernel that does (almost) nothing, by grid size and LDS footprint -
// the floor under every small-batch kernel time. hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_BYTES, int MODE>
__global__ __launch_bounds__(256) void k_floor(float* out, const float* in, int spin)
{
    __shared__ float s[LDS_BYTES / 4];
    const int tid = threadIdx.x;
    if (MODE == 0) {
        if (spin == 12345) s[tid] = 1.0f;   // keep the LDS allocation alive
        if (spin == 12345) out[blockIdx.x * 256 + tid] = s[(tid + 1) & 255];
    } else if (MODE == 1) {   // one dependent HBM round trip per thread
        const float v = in[(size_t)blockIdx.x * 256 + tid];
        s[tid] = v;
        __syncthreads();
        out[(size_t)blockIdx.x * 256 + tid] = s[(tid + 1) & 255];
    } else {   // `spin` dependent FMAs per thread: pure VALU time
        float a = in[tid];
        for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
        out[(size_t)blockIdx.x * 256 + tid] = a;
    }
}

template <int LDS_BYTES, int MODE>
void run(const char* name, int grid, int spin, float* out, float* in)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    for (int it = 0; it < 12; ++it) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_floor<LDS_BYTES, MODE>), dim3(grid), dim3(256), 0, 0, out, in, spin);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-44s grid=%5d lds=%6d spin=%5d  event span: avg %7.2f us, best %7.2f us\n", name, grid, LDS_BYTES, spin, sum / 10 * 1e3, best * 1e3);
}

int main()
{
    float *out, *in;
    (void)hipMalloc(&out, 8192 * 256 * sizeof(float));
    (void)hipMalloc(&in, 8192 * 256 * sizeof(float));
    (void)hipMemset(in, 0, 8192 * 256 * sizeof(float));
    for (int grid : {256, 768, 1024, 2048, 4096}) {
        run<1024, 0>("empty", grid, 0, out, in);
        run<40448, 0>("empty, 40 KB LDS", grid, 0, out, in);
        run<1024, 1>("one HBM round trip", grid, 0, out, in);
        run<40448, 1>("one HBM round trip, 40 KB LDS", grid, 0, out, in);
        run<1024, 2>("VALU spin", grid, 2000, out, in);
        run<40448, 2>("VALU spin, 40 KB LDS", grid, 2000, out, in);
    }
    return 0;
}
