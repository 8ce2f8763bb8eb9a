// Fills every CU's LDS with a float pattern for N seconds (a second process beside the encoder: what an uninitialised LDS read would pick up
// then is this pattern instead of the encoder's own leftovers). usage: lds_noise <seconds> <float value>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_fill(float v, int iters)
{
    extern __shared__ float s[];
    for (int it = 0; it < iters; ++it)
        for (int i = threadIdx.x; i < 15 * 1024; i += blockDim.x) s[i] = v + (float)it * 0.0f;
    __syncthreads();
    if (s[threadIdx.x] == 12345.0f) printf("x");
}
int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 10.0;
    const float v = argc > 2 ? (float)atof(argv[2]) : 1.0e30f;
    hipFuncSetAttribute((const void*)k_fill, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 60 * 1024, 0, v, 4);
        if ((++n & 63) == 0) hipDeviceSynchronize();
    }
    hipDeviceSynchronize();
    printf("lds_noise: %ld launches\n", n);
    return 0;
}
