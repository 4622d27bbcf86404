// How large may a kernel's by-value arguments be?  (the multi-slot launch of scan_flat_kernel passes its batch table as
// kernel arguments).  Build: hipcc --offload-arch=gfx950 -O2 kernarg_size.hip -o kernarg_size
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Blob { int v[N]; };
template <int N> __global__ void k(Blob<N> b, int *out) { if (threadIdx.x == 0) out[0] = b.v[N - 1] + b.v[0]; }
template <int N> void run(int *d)
{
    Blob<N> b;
    for (int i = 0; i < N; ++i) b.v[i] = i;
    hipMemset(d, 0, 4);
    k<N><<<1, 64>>>(b, d);
    hipError_t e = hipGetLastError();
    hipError_t s = hipDeviceSynchronize();
    int h = -1;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%6d bytes: launch %s, sync %s, result %d (expect %d)\n", 4 * N, hipGetErrorName(e), hipGetErrorName(s), h, N - 1);
}
int main()
{
    int *d;
    hipMalloc(&d, 4);
    run<512>(d); run<1000>(d); run<1020>(d); run<1536>(d); run<2048>(d); run<4096>(d); run<8192>(d);
    return 0;
}
