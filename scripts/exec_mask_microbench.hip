// exec_mask_microbench.hip — does a wave64 VALU instruction get cheaper when half of its lanes are off? (tooling, gfx950)
//   hipcc -O3 --offload-arch=gfx950 scripts/exec_mask_microbench.hip -o scripts/exec_mask_microbench && scripts/exec_mask_microbench
// 5 waves per SIMD, each a loop of 16 independent v_fma_f32 chains; only the lanes selected by `mask` run the loop.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(256) void loop(unsigned long long mask, uint32_t iters, float *out, unsigned long long *cycles, unsigned long long *real, uint32_t mixed)
{
    const uint32_t lane = threadIdx.x & 63u;
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
    const unsigned long long r0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
    // `mixed`: only every fifth workgroup (one of the five waves of each SIMD) runs with the thin mask, the others with all lanes
    const unsigned long long mine = (mixed && blockIdx.x % 5u != 0u) ? ~0ull : mask;
    if ((mine >> lane) & 1ull)
    {
        for (uint32_t it = 0; it < iters; ++it)
        {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(1.0001f));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0)
    {
        cycles[blockIdx.x] = t1 - t0;
        real[blockIdx.x] = r1 - r0;
    }
}

int main()
{
    const uint32_t blocks = 256 * 5, iters = 4096;
    float *out;
    unsigned long long *cyc, *real;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 8);
    hipMalloc(&real, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct
    {
        const char *name;
        unsigned long long mask;
    } cases[] = {{"all 64 lanes", ~0ull},           {"lanes 0-31", 0xFFFFFFFFull},          {"lanes 32-63", 0xFFFFFFFF00000000ull},
                 {"even lanes", 0x5555555555555555ull}, {"lanes 0-15", 0xFFFFull},               {"lane 0", 1ull},
                 {"lanes 0-15 + 32-47", 0x0000FFFF0000FFFFull}, {"lanes 0-47", 0xFFFFFFFFFFFFull}, {"lanes 0-23", 0xFFFFFFull},
                 {"every 4th lane (16)", 0x1111111111111111ull}, {"lanes 0-7", 0xFFull}, {"every 8th lane (8)", 0x0101010101010101ull},
                 {"lanes 0-8 (9)", 0x1FFull}, {"lanes 0-9 (10)", 0x3FFull}, {"lanes 0-11 (12)", 0xFFFull}, {"lanes 0-13 (14)", 0x3FFFull}, {"lanes 0-16 (17)", 0x1FFFFull}, {"lanes 0-19 (20)", 0xFFFFFull}, {"9 lanes spread", 0x0101010101010103ull}, {"lanes 0-3", 0xFull}, {"lanes 0-1", 0x3ull}, {"lanes 0 and 32", 0x100000001ull}, {"lanes 0,16,32,48", 0x0001000100010001ull}};
    for (uint32_t mixed = 0; mixed < 2; ++mixed)
    for (auto &c : cases)
    {
        if (mixed && __builtin_popcountll(c.mask) > 16) continue;
        hipLaunchKernelGGL(loop, dim3(blocks), dim3(256), 0, 0, c.mask, 64u, out, cyc, real, mixed);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop, dim3(blocks), dim3(256), 0, 0, c.mask, iters, out, cyc, real, mixed);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)blocks * 4 * iters * 16; // wave-instructions
        static unsigned long long hc[256 * 5], hr[256 * 5];
        hipMemcpy(hc, cyc, blocks * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hr, real, blocks * 8, hipMemcpyDeviceToHost);
        double sc = 0, sr = 0;
        for (uint32_t b = 0; b < blocks; ++b)
        {
            sc += (double)hc[b];
            sr += (double)hr[b];
        }
        // per wave: iters * 16 instructions; 5 waves share a SIMD
        printf("%s%-22s %7.3f ms   %.1f G wave-inst/s   s_memtime ticks per instruction and wave %.2f   100 MHz ticks per wave %.0f  => %.2f GHz if s_memtime is the core clock\n",
               mixed ? "1 thin wave of 5: " : "", c.name, ms, insts / (ms * 1e6), sc / blocks / (iters * 16.0), sr / blocks, (sc / blocks) / (sr / blocks) * 0.1);
    }
    return 0;
}
