// Per-CU store bandwidth vs number of storing CUs (gfx950): does a CU drain its output faster when fewer CUs store at once?
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/store_bw tools/micro/store_bw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void store_kernel(unsigned short* out, long stride_elems, int reps, int mode) {
    // each workgroup writes `reps` tiles of 256 rows x 256 bf16 (row pitch 4352 elements) like the v8 epilogue: 8 B per lane, 16 lanes per row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    for (int r = 0; r < reps; ++r) {
        unsigned short* base = out + ((long)blockIdx.x * reps + r) * stride_elems;
        for (int mi = 0; mi < 8; ++mi)
            for (int q = 0; q < 4; ++q) {
                const int row = wr * 128 + mi * 16 + q * 4 + (lane >> 4);
                const int col = wc * 64 + (lane & 15) * 4;
                uint2 v = make_uint2(tid + r, mi + q);
                *reinterpret_cast<uint2*>(base + (long)row * 4352 + col) = v;
            }
        if (mode == 1) __builtin_amdgcn_s_sleep(127);
    }
}

int main() {
    const long tile_span = 256L * 4352;      // elements between consecutive tiles of one workgroup (distinct memory)
    const int reps = 64;
    unsigned short* d;
    const long total = 256L * reps * tile_span;
    if (hipMalloc(&d, total * 2) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int ncu : {8, 16, 32, 64, 128, 256}) {
        store_kernel<<<ncu, 512>>>(d, tile_span, reps, 0);
        hipDeviceSynchronize();
        hipEventRecord(a);
        store_kernel<<<ncu, 512>>>(d, tile_span, reps, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)ncu * reps * 256 * 256 * 2;
        printf("workgroups %3d: %.3f ms, %.1f GB/s per CU, %.2f TB/s total, %.2f us per 128 KiB tile\n", ncu, ms, bytes / ncu / ms / 1e6,
               bytes / ms / 1e9, ms * 1e3 / reps);
    }
    return 0;
}
