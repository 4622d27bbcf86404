// Micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 with 1 / 2 / 4 independent accumulators per wave,
// one wave per SIMD (what a small conv launch gives the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = float __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float *out, float a, float b, int n)
{
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = f32x16{0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + t, b + u, acc[t], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < NACC; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (float)(t1 - t0);
}

template <int NACC>
void run(float *d, int waves_per_simd, int blocks = 256, int n = 256)
{
    k<NACC><<<blocks, 256 * waves_per_simd>>>(d, 1.0f, 0.5f, n);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NACC><<<blocks, 256 * waves_per_simd>>>(d, 1.0f, 0.5f, n);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc;
    hipMemcpy(&cyc, d + blocks * 256 * waves_per_simd, sizeof(float), hipMemcpyDeviceToHost);
    const double mf = (double)n * 8 * NACC;
    printf("accumulators %d, waves/SIMD %d, %3d workgroups, %4d MFMAs per wave: %.1f memtime ticks per MFMA, %.2f ns per MFMA (kernel %.1f us)\n",
           NACC, waves_per_simd, blocks, n * 8 * NACC, cyc / mf, ms * 1e6 / mf, ms * 1e3);
}

int main()
{
    float *d;
    hipMalloc(&d, (256 * 1024 + 8) * sizeof(float));
    run<1>(d, 1); run<2>(d, 1); run<4>(d, 1);
    run<1>(d, 2); run<1>(d, 4);
    // the shape of a small conv launch: a few hundred MFMAs per wave, fewer workgroups than CUs, launched cold
    run<1>(d, 1, 100, 48); run<1>(d, 1, 200, 96); run<1>(d, 1, 200, 24);
    return 0;
}
