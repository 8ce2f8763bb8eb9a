// Development aid: prints what v_permlane32_swap / v_permlane16_swap do on this device (the SIMT harness of tools/emu restates
// them; this is the check of that restatement on hardware).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o)
{
    const int l = threadIdx.x;
    const auto r = __builtin_amdgcn_permlane32_swap(1000 + l, 2000 + l, false, false);
    const auto q = __builtin_amdgcn_permlane16_swap(1000 + l, 2000 + l, false, false);
    o[l] = r[0]; o[64 + l] = r[1]; o[128 + l] = q[0]; o[192 + l] = q[1];
}
int main()
{
    int* d; int h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int e32a = l < 32 ? 1000 + l : 2000 + (l - 32), e32b = l < 32 ? 1000 + (l + 32) : 2000 + l;
        const int e16a = (l & 16) == 0 ? 1000 + l : 2000 + (l ^ 16), e16b = (l & 16) == 0 ? 1000 + (l ^ 16) : 2000 + l;
        bad += (h[l] != e32a) + (h[64 + l] != e32b) + (h[128 + l] != e16a) + (h[192 + l] != e16b);
    }
    printf("permlane swaps match the restated semantics: %s (%d differences)\n", bad ? "NO" : "yes", bad);
    for (int l = 0; l < 64; l += 8) printf("lane %2d: swap32 -> (%d, %d) swap16 -> (%d, %d)\n", l, h[l], h[64 + l], h[128 + l], h[192 + l]);
    return bad != 0;
}
