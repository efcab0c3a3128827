// How many 256-thread workgroups with X bytes of dynamic LDS does a CU of gfx950 really hold?  Every workgroup spins for a fixed wall time;
// 2 x CUs workgroups finish in one spin period iff two are co-resident per CU.  (Round 6: gemm_nt_bf16_v11_kernel declares exactly half of the
// 160 KiB.)   build: hipcc --offload-arch=gfx950 -O2 tools/micro/lds_occupancy_probe.hip -o gpurun_out/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256, 2) void spin(unsigned long long ticks, int* sink) {
    extern __shared__ char smem[];
    smem[threadIdx.x] = (char)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (smem[(threadIdx.x + 1) & 255] == 77 && sink) *sink = 1;
}
int main() {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int sizes[] = {65536, 73728, 80000, 81920 - 1280, 81920, 98304};
    for (int lds : sizes) {
        hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int occ = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 256, lds);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(2 * ncu), dim3(256), lds, 0, 10000ull /* 100 us at 100 MHz */, (int*)nullptr);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("dynamic LDS %6d B: occupancy API says %d blocks/CU; %d workgroups of 100 us took %.1f us -> %s\n", lds, occ, 2 * ncu, ms * 1e3,
               ms < 0.15 ? "2 per CU" : "ONE per CU");
    }
    return 0;
}
