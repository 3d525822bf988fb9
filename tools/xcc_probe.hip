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

int main() {
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
