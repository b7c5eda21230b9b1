// Probe: v_mfma_f64_16x16x4_f64 with A = B = a 4-row slab of M (16 columns) must give M^T M.  hipcc --offload-arch=gfx950 -o mfma_f64_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* M, int rows, double* D)
{
    const int lane = threadIdx.x, col = lane & 15, kr = lane >> 4;
    f64x4 acc = {0, 0, 0, 0};
    for (int c = 0; c < rows / 4; c++) {
        const double m = M[(4 * c + kr) * 16 + col];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0);
    }
    for (int reg = 0; reg < 4; reg++) D[((lane >> 4) + 4 * reg) * 16 + (lane & 15)] = acc[reg];
}
int main()
{
    const int rows = 32;
    double hM[rows * 16], hD[256], ref[256];
    for (int i = 0; i < rows * 16; i++) hM[i] = (double)((i * 7 + (i / 16) * 3) % 11 - 5);
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int r = 0; r < rows; r++) s += hM[r*16 + i] * hM[r*16 + j]; ref[i*16 + j] = s; }
    double *dM, *dD;
    hipMalloc(&dM, sizeof(hM)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dM, hM, sizeof(hM), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dM, rows, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) bad += hD[i] != ref[i];
    printf("mismatches: %d of 256 (D[1][2] = %g, ref %g)\n", bad, hD[18], ref[18]);
    return bad != 0;
}
