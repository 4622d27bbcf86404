// Micro-benchmark: issue cost of float64-related VALU instructions on gfx950.
// One wave per SIMD (256 threads/block, 1 block per CU), 8 independent chains,
// reports cycles per wave-instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define ITERS 512

template <int OP>
__global__ void k(double *out, double seed, int n)
{
    double v[CHAINS];
    float f[CHAINS];
    int iv[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { v[c] = seed + c + threadIdx.x * 1e-3; f[c] = (float)v[c]; iv[c] = threadIdx.x + c; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) v[c] = __builtin_fma(v[c], 1.0000001, 0.5);
            if (OP == 1) v[c] = v[c] + 1.5;
            if (OP == 2) v[c] = v[c] * 1.0000001;
            if (OP == 3) { iv[c] = (int)v[c]; v[c] = v[c] + (double)0; asm volatile("" : "+v"(iv[c])); v[c] = __hiloint2double(__double2hiint(v[c]), __double2loint(v[c]) ^ iv[c]); }
            if (OP == 4) { v[c] = (double)iv[c]; asm volatile("" : "+v"(v[c])); iv[c] ^= __double2loint(v[c]); }
            if (OP == 5) { v[c] = (double)f[c]; asm volatile("" : "+v"(v[c])); f[c] = __int_as_float(__float_as_int(f[c]) ^ (__double2hiint(v[c]) & 1)); }
            if (OP == 6) { f[c] = (float)v[c]; asm volatile("" : "+v"(f[c])); v[c] = __hiloint2double(__double2hiint(v[c]), __double2loint(v[c]) ^ (__float_as_int(f[c]) & 1)); }
            if (OP == 7) v[c] = __builtin_amdgcn_fract(v[c]) + 1.25;  // fract + add
            if (OP == 8) v[c] = __builtin_floor(v[c]) + 0.37;          // floor + add
            if (OP == 9) v[c] = (v[c] < 3.0) ? 3.0 : v[c] * 0.99;      // cmp + cndmask x2 + mul
            if (OP == 10) v[c] = __builtin_fmin(v[c], 1e300) * 1.01;   // min + mul
            if (OP == 11) f[c] = __builtin_fmaf(f[c], 1.0001f, 0.5f);  // f32 fma reference
            if (OP == 12) v[c] = __builtin_rint(v[c]) + 0.37;          // rndne + add
            if (OP == 13) v[c] = __builtin_sqrt(v[c] * v[c] + 1.0);    // sqrt expansion
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c] + f[c] + iv[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

template <int OP>
void run(const char *name, double *d, int extra_ops)
{
    const int blocks = 256, threads = 256;
    k<OP><<<blocks, threads>>>(d, 1.0, ITERS);
    hipDeviceSynchronize();
    k<OP><<<blocks, threads>>>(d, 1.0, ITERS);
    hipDeviceSynchronize();
    double cyc;
    hipMemcpy(&cyc, d + blocks * threads, sizeof(double), hipMemcpyDeviceToHost);
    printf("%-28s %7.2f cycles per (chain step) per wave  [step = %d instr]\n", name, cyc / (ITERS * CHAINS), extra_ops);
}

int main()
{
    double *d;
    hipMalloc(&d, (256 * 256 + 8) * sizeof(double));
    run<0>("v_fma_f64", d, 1);
    run<1>("v_add_f64", d, 1);
    run<2>("v_mul_f64", d, 1);
    run<3>("cvt_i32_f64 (+add,xor)", d, 3);
    run<4>("cvt_f64_i32 (+xor)", d, 2);
    run<5>("cvt_f64_f32 (+and,xor)", d, 3);
    run<6>("cvt_f32_f64 (+and,xor)", d, 3);
    run<7>("fract_f64 + add", d, 2);
    run<8>("floor_f64 + add", d, 2);
    run<9>("cmp_f64+2cndmask+mul", d, 4);
    run<10>("min_f64 + mul", d, 2);
    run<11>("v_fma_f32", d, 1);
    run<12>("rndne_f64 + add", d, 2);
    run<13>("sqrt(f64) expansion + fma", d, 0);
    return 0;
}
