// issue_rate_probe.hip -- issue cost (cycles per wave-instruction on one SIMD) of the VALU / LDS instructions the Hessian
// kernel is made of (surf.hip: det_layer_c): int add, i32 -> f32, f32 mul (plain and packed), f32 -> f64, f64 add, f64 -> f32.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/issue_rate_probe.hip -o tools/probe/issue_rate_probe && tools/probe/issue_rate_probe
// One workgroup of `waves` waves on one CU (waves = 1: one wave alone on its SIMD; 4: one per SIMD; 8 / 16: two / four per SIMD),
// 64 independent instructions per loop iteration (eight register chains), s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

enum Op { ADD_U32, ADD3_U32, CVT_F32_I32, MUL_F32, PK_MUL_F32, ADD_F32, CVT_F64_F32, ADD_F64, CVT_F32_F64, FMA_F64, MUL_F64, CVT_F64_I32, LDS_READ_B32, LDS_READ2_B32, SAD_U32, FMA_F32, PK_FMA_F32, SUB_U32, N_OPS };
static const char* kNames[N_OPS] = { "v_add_u32", "v_add3_u32", "v_cvt_f32_i32", "v_mul_f32", "v_pk_mul_f32 (2 products)", "v_add_f32", "v_cvt_f64_f32", "v_add_f64",
                                      "v_cvt_f32_f64", "v_fma_f64", "v_mul_f64", "v_cvt_f64_i32", "ds_read_b32", "ds_read2_b32", "v_sad_u32", "v_fma_f32", "v_pk_fma_f32 (2 fmas)", "v_sub_u32" };

template <int OP>
__global__ void k_rate(long long* out, int iters, float seed)
{
    __shared__ int lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    float f[8]; double d[8]; int n[8]; float2 p[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { f[k] = seed + k; d[k] = seed * 3 + k; n[k] = (int)seed + k + threadIdx.x; p[k] = make_float2(seed + k, seed - k); }
    const int lane_addr = (threadIdx.x & 63) * 4;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
                else if constexpr (OP == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
                else if constexpr (OP == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[k]) : "v"(n[k]));
                else if constexpr (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(seed));
                else if constexpr (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[k]));
                else if constexpr (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(seed));
                else if constexpr (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(f[k]));
                else if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(d[(k + 1) & 7]));
                else if constexpr (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(d[k]));
                else if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[k]) : "v"(d[(k + 1) & 7]));
                else if constexpr (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(d[(k + 1) & 7]));
                else if constexpr (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[k]) : "v"(n[k]));
                else if constexpr (OP == SAD_U32) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(n[k]) : "v"(n[(k + 1) & 7]), "s"(0x4B000000));
                else if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k]) : "v"(seed));
                else if constexpr (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[k]));
                else if constexpr (OP == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
                else if constexpr (OP == LDS_READ_B32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(n[k]) : "v"(lane_addr), "n"(k * 256));
                else if constexpr (OP == LDS_READ2_B32) { long long v; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(lane_addr), "n"(k * 2), "n"(k * 2 + 64)); n[k] = (int)v; }
            }
        }
        if constexpr (OP == LDS_READ_B32 || OP == LDS_READ2_B32) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0; double dacc = 0; int nacc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { acc += f[k] + p[k].x + p[k].y; dacc += d[k]; nacc += n[k]; }
    if (acc == 1.2345f && dacc == 1.25 && nacc == 77) out[1] = 1;          // keep the chains alive
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <int OP>
static int run(long long* d_out, int waves)
{
    const int iters = 2000;
    long long h = 0;
    hipLaunchKernelGGL(k_rate<OP>, dim3(1), dim3(64 * waves), 0, 0, d_out, iters, 1.5f);
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_rate<OP>, dim3(1), dim3(64 * waves), 0, 0, d_out, iters, 1.5f);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    // s_memtime ticks at a constant 100 MHz on gfx9; report both raw ticks per instruction and the ratio to v_add_f32 of the same run
    std::printf("%-28s waves %2d: %8.4f ticks / wave-instruction\n", kNames[OP], waves, (double)h / (iters * 64.0));
    return 0;
}

int main()
{
    long long* d_out = nullptr;
    CHK(hipMalloc(&d_out, 16));
    // (one wave only: with several waves per SIMD the oldest wave issues first and thread 0's clock shows no contention)
    for (int waves : { 1 }) {
        if (run<ADD_F32>(d_out, waves)) return 1;
        if (run<ADD_U32>(d_out, waves)) return 1;
        if (run<ADD3_U32>(d_out, waves)) return 1;
        if (run<CVT_F32_I32>(d_out, waves)) return 1;
        if (run<MUL_F32>(d_out, waves)) return 1;
        if (run<PK_MUL_F32>(d_out, waves)) return 1;
        if (run<CVT_F64_F32>(d_out, waves)) return 1;
        if (run<ADD_F64>(d_out, waves)) return 1;
        if (run<CVT_F32_F64>(d_out, waves)) return 1;
        if (run<FMA_F64>(d_out, waves)) return 1;
        if (run<MUL_F64>(d_out, waves)) return 1;
        if (run<CVT_F64_I32>(d_out, waves)) return 1;
        if (run<LDS_READ_B32>(d_out, waves)) return 1;
        if (run<LDS_READ2_B32>(d_out, waves)) return 1;
        if (run<SAD_U32>(d_out, waves)) return 1;
        if (run<FMA_F32>(d_out, waves)) return 1;
        if (run<PK_FMA_F32>(d_out, waves)) return 1;
        if (run<SUB_U32>(d_out, waves)) return 1;
    }
    return 0;
}
