// tools/xcc_probe.hip — where does the dispatcher put workgroup b?  Prints, for a 2048-block
// launch of 256-thread blocks, how many distinct XCC ids the blocks with equal (b % 8) see and
// the XCC id of the first 32 blocks.  Build/run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void probe(int* xcc, int* cu) {
    if (threadIdx.x == 0) {
        unsigned x, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        xcc[blockIdx.x] = (int)(x & 0xf);
        cu[blockIdx.x] = (int)hw;
    }
    // keep the block alive a little so that all blocks are co-resident
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}

// Two-dimensional launches (the strip schedule of the sweeps, csrc/pi_sweep_kernels.hip): with gridDim.x a multiple of 8,
// is workgroup (bx, by) on XCD bx % 8 for every by?  1 024-thread workgroups, more of them than fit the chip at once.
__global__ void probe2d(int* xcc) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc[blockIdx.y * gridDim.x + blockIdx.x] = (int)(x & 0xf);
    }
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(10);
}
static void two_dimensional(int gx, int gy, int early_exit_from) {
    int* d;
    hipMalloc(&d, gx * gy * sizeof(int));
    hipMemset(d, 0xff, gx * gy * sizeof(int));
    (void)early_exit_from;
    hipLaunchKernelGGL(probe2d, dim3(gx, gy), dim3(1024), 0, 0, d);
    std::vector<int> x(gx * gy);
    hipMemcpy(x.data(), d, gx * gy * sizeof(int), hipMemcpyDeviceToHost);
    std::set<int> first_row;
    for (int b = 0; b < 8; ++b) first_row.insert(x[b]);
    int mismatches = 0;
    for (int y = 0; y < gy; ++y)
        for (int b = 0; b < gx; ++b)
            if (x[y * gx + b] != x[b % 8]) ++mismatches;
    printf("2-D launch %d x %d of 1024-thread workgroups: XCDs of the first 8: %zu distinct; workgroups not on the XCD of (bx %% 8, 0): %d of %d\n",
           gx, gy, first_row.size(), mismatches, gx * gy);
    hipFree(d);
}

int main() {
    two_dimensional(256, 80, 0);       // the 80^4 evaluation sweep's grid
    two_dimensional(1000, 80, 0);      // ... the improvement sweep's
    two_dimensional(96, 625, 0);       // a 25^6 launch with a sub-plane as the period
    two_dimensional(24, 7, 0);
    const int nb = 2048;
    int *d_x, *d_c;
    hipMalloc(&d_x, nb * sizeof(int));
    hipMalloc(&d_c, nb * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d_x, d_c);
    std::vector<int> x(nb), c(nb);
    hipMemcpy(x.data(), d_x, nb * sizeof(int), hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d_c, nb * sizeof(int), hipMemcpyDeviceToHost);
    printf("first 32 blocks -> xcc:");
    for (int b = 0; b < 32; ++b) printf(" %d", x[b]);
    printf("\n");
    for (int r = 0; r < 8; ++r) {
        std::set<int> s;
        for (int b = r; b < nb; b += 8) s.insert(x[b]);
        printf("blocks with b%%8==%d see %zu distinct XCC ids:", r, s.size());
        for (int v : s) printf(" %d", v);
        printf("\n");
    }
    int per[16] = {0};
    for (int b = 0; b < nb; ++b) per[x[b]]++;
    printf("blocks per xcc:");
    for (int i = 0; i < 8; ++i) printf(" %d", per[i]);
    printf("\n");
    return 0;
}
