// Semantics probe of the GFX9 wave-wide DPP shifts on gfx950: which lane does lane i read with wave_shr:1 / wave_shl:1,
// and what do the lanes without a source get (old value, bound_ctrl = 0)?
// Build: hipcc --offload-arch=gfx950 -O3 -o dpp_wave_shift tools/ubench/dpp_wave_shift.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, float *out)
{
    const float v = in[threadIdx.x];
    const float r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1.0f), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
    const float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1.0f), __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
    out[threadIdx.x] = r;
    out[64 + threadIdx.x] = l;
}
int main()
{
    float h[64], o[128], *d, *e;
    for (int i = 0; i < 64; ++i) h[i] = (float)i;
    hipMalloc(&d, sizeof h); hipMalloc(&e, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, e);
    hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
    printf("wave_shr:1  lane0 %g lane1 %g lane16 %g lane32 %g lane63 %g\n", o[0], o[1], o[16], o[32], o[63]);
    printf("wave_shl:1  lane0 %g lane15 %g lane31 %g lane62 %g lane63 %g\n", o[64], o[64 + 15], o[64 + 31], o[64 + 62], o[64 + 63]);
    return 0;
}
