// Development aid: prints the device properties the host layer's launch heuristics read.
#include <hip/hip_runtime.h>
#include <cstdio>
int main()
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    printf("name %s arch %s CUs %d clockRate %d kHz LDS/block %zu LDS/CU %zu regs/block %d L2 %d wave %d maxThreads/CU %d\n", p.name, p.gcnArchName,
           p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.l2CacheSize, p.warpSize,
           p.maxThreadsPerMultiProcessor);
    return 0;
}
