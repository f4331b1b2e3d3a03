// how many cycles does one posvx_rows take a wave that is alone on its SIMD, and with 2 waves a SIMD? (probe)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o tmp_ab/posvx_cpi scripts/probes/posvx_cpi.hip
#include "../../mir_optim_amd/csrc/batched_kernel.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mirlsq;
int main()
{
    const int count = 4 * 2048 * 16;       // systems
    std::vector<float> P(64 * (size_t)count), b(8 * (size_t)count);
    srand(3);
    for (int p = 0; p < count; ++p) {
        float G[8][8];
        for (auto& row : G) for (int j = 0; j < 8; ++j) row[j] = 2.0f * rand() / (float)RAND_MAX - 1.0f;
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { float v = (i == j) ? 0.05f : 0.0f; for (int k = 0; k < 8; ++k) v += G[k][i] * G[k][j]; P[(size_t)p * 64 + i * 8 + j] = v; }
        for (int i = 0; i < 8; ++i) b[(size_t)p * 8 + i] = 2.0f * rand() / (float)RAND_MAX - 1.0f;
    }
    float *dP, *db, *dx; int* di;
    (void)hipMalloc(&dP, 4 * P.size()); (void)hipMalloc(&db, 4 * b.size()); (void)hipMalloc(&dx, 4 * b.size()); (void)hipMalloc(&di, 4 * count);
    (void)hipMemcpy(dP, P.data(), 4 * P.size(), hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 4 * b.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {256, 1024, 2048, 4096, 8192}) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            k_posvx_rows<8><<<blocks, 64>>>(dP, db, count, dx, di);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            const double calls_per_wave = (double)count / 4 / blocks;
            if (rep) printf("%5d waves (%.1f per SIMD): %.3f ms, %.0f solve calls per wave, %.2f us = %.0f cycles (2.4 GHz) per call\n", blocks, blocks / 1024.0, ms,
                            calls_per_wave, 1e3 * ms / calls_per_wave, 2.4e3 * 1e3 * ms / calls_per_wave / 1e3);
        }
    }
    return 0;
}
