// permlane_probe.hip -- what v_permlane16_swap / v_permlane32_swap (gfx950) do, as used by group_pick in solve_lds.h:
// register t of lane group g holds 100 t + 10 g + (lane & 15) % 10; after group_pick<JB> lane group k must hold register k of group JB.
// hipcc --offload-arch=gfx950 -O2 -o permlane_probe permlane_probe.hip && ./permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mir_optim_amd/csrc/common.h"
using namespace mirlsq;

template <int JB> __device__ void one(int* out, int lane)
{
    const int g = lane >> 4;
    int r[4];
    for (int t = 0; t < 4; ++t) r[t] = 1000 * t + 100 * g + (lane & 15);
    out[JB * 64 + lane] = group_pick<JB>(r[0], r[1], r[2], r[3]);
}
__global__ void k(int* out)
{
    const int lane = threadIdx.x;
    one<0>(out, lane); one<1>(out, lane); one<2>(out, lane); one<3>(out, lane);
}
int main()
{
    int* d; int h[256];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int jb = 0; jb < 4; ++jb)
        for (int l = 0; l < 64; ++l) {
            const int want = 1000 * (l >> 4) + 100 * jb + (l & 15);
            if (h[jb * 64 + l] != want) { if (bad < 8) printf("JB %d lane %d: got %d want %d\n", jb, l, h[jb * 64 + l], want); ++bad; }
        }
    printf("permlane group_pick: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return bad != 0;
}
