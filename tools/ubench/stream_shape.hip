// Micro-benchmark: the floor for the I/O shape of scan_preprocess (no arithmetic).
// Per scan (N = 450): read 4N, write 8N (flow) + 8N (target_cls i64) + 8N (target_reg) + 4N (mask).
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_shape stream_shape.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int N = 450;

// SPB samples per workgroup, 2 points per lane (225 active lanes per sample)
template <int SPB, bool NT>
__global__ __launch_bounds__(256) void shape_kernel(const float *r, float *flow, long long *cls, float *reg, float *mask, int B)
{
    for (int s = 0; s < SPB; ++s) {
        const int b = blockIdx.x * SPB + s;
        if (b >= B) return;
        const int p = threadIdx.x;  // pair index
        if (p >= N / 2) continue;
        const float2 v = reinterpret_cast<const float2 *>(r + (long long)b * N)[p];
        const float4 f = make_float4(v.x, v.y, v.x + 1.f, v.y + 1.f);
        float4 *fo = reinterpret_cast<float4 *>(flow + (long long)b * N * 2) + p;
        float4 *ro = reinterpret_cast<float4 *>(reg + (long long)b * N * 2) + p;
        longlong2 *co = reinterpret_cast<longlong2 *>(cls + (long long)b * N) + p;
        float2 *mo = reinterpret_cast<float2 *>(mask + (long long)b * N) + p;
        longlong2 c; c.x = v.x > 3.f; c.y = v.y > 3.f;
        if (NT) {
            __builtin_nontemporal_store(f.x, &fo->x); __builtin_nontemporal_store(f.y, &fo->y);
            __builtin_nontemporal_store(f.z, &fo->z); __builtin_nontemporal_store(f.w, &fo->w);
            __builtin_nontemporal_store(f.x, &ro->x); __builtin_nontemporal_store(f.y, &ro->y);
            __builtin_nontemporal_store(f.z, &ro->z); __builtin_nontemporal_store(f.w, &ro->w);
            __builtin_nontemporal_store(c.x, &co->x); __builtin_nontemporal_store(c.y, &co->y);
            __builtin_nontemporal_store(v.x, &mo->x); __builtin_nontemporal_store(v.y, &mo->y);
        } else {
            *fo = f; *ro = f; *co = c; *mo = v;
        }
    }
}

// 4 points per lane (113 active lanes of a 128-thread workgroup), streaming stores
__global__ __launch_bounds__(128) void shape4_kernel(const float *r, float *flow, long long *cls, float *reg, float *mask, int B)
{
    const int b = blockIdx.x;
    const int p = threadIdx.x;            // group of 4 points
    if (4 * p >= N) return;
    const bool full = 4 * p + 3 < N;      // N = 450: the last group has 2 points
    using F4 = float __attribute__((ext_vector_type(4)));
    using L2 = long long __attribute__((ext_vector_type(2)));
    const float *row = r + (long long)b * N + 4 * p;
    const float v0 = row[0], v1 = row[1], v2 = full ? row[2] : 0.f, v3 = full ? row[3] : 0.f;
    F4 *fo = reinterpret_cast<F4 *>(flow + (long long)b * N * 2) + 2 * p;
    F4 *ro = reinterpret_cast<F4 *>(reg + (long long)b * N * 2) + 2 * p;
    L2 *co = reinterpret_cast<L2 *>(cls + (long long)b * N) + 2 * p;
    float *mo = mask + (long long)b * N + 4 * p;
    __builtin_nontemporal_store(F4{v0, v1, v0 + 1.f, v1 + 1.f}, fo);
    __builtin_nontemporal_store(F4{v0, v1, v0 + 1.f, v1 + 1.f}, ro);
    __builtin_nontemporal_store(L2{v0 > 3.f, v1 > 3.f}, co);
    using F2 = float __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(F2{v0, v1}, reinterpret_cast<F2 *>(mo));
    if (full) {
        __builtin_nontemporal_store(F4{v2, v3, v2 + 1.f, v3 + 1.f}, fo + 1);
        __builtin_nontemporal_store(F4{v2, v3, v2 + 1.f, v3 + 1.f}, ro + 1);
        __builtin_nontemporal_store(L2{v2 > 3.f, v3 > 3.f}, co + 1);
        __builtin_nontemporal_store(F2{v2, v3}, reinterpret_cast<F2 *>(mo) + 1);
    }
}

// flat grid-stride copy-like kernel: same bytes, ideal access pattern
__global__ __launch_bounds__(256) void flat_kernel(const float4 *in, float4 *out, long long n_in, long long n_out)
{
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    float4 acc = make_float4(0, 0, 0, 0);
    for (long long i = tid; i < n_in; i += stride) { float4 v = in[i]; acc.x += v.x; acc.y += v.y; }
    for (long long i = tid; i < n_out; i += stride) out[i] = acc;
}

int main()
{
    const int B = 4096, RING = 8, ITERS = 400;
    std::vector<float *> r(RING), flow(RING), reg(RING), mask(RING);
    std::vector<long long *> cls(RING);
    for (int i = 0; i < RING; ++i) {
        CK(hipMalloc(&r[i], (size_t)B * N * 4)); CK(hipMemset(r[i], 0, (size_t)B * N * 4));
        CK(hipMalloc(&flow[i], (size_t)B * N * 8)); CK(hipMalloc(&reg[i], (size_t)B * N * 8));
        CK(hipMalloc(&cls[i], (size_t)B * N * 8)); CK(hipMalloc(&mask[i], (size_t)B * N * 4));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)B * N * 32;
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 20; ++i) launch(i % RING);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < ITERS; ++i) launch(i % RING);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %7.2f us/launch  %7.0f GB/s\n", name, ms / ITERS * 1e3, bytes / (ms / ITERS * 1e-3) / 1e9);
        fflush(stdout);
    };
    run("shape SPB=1", [&](int k) { shape_kernel<1, false><<<B, 256>>>(r[k], flow[k], cls[k], reg[k], mask[k], B); });
    run("shape SPB=2", [&](int k) { shape_kernel<2, false><<<B / 2, 256>>>(r[k], flow[k], cls[k], reg[k], mask[k], B); });
    run("shape SPB=4", [&](int k) { shape_kernel<4, false><<<B / 4, 256>>>(r[k], flow[k], cls[k], reg[k], mask[k], B); });
    run("shape 4 pts/lane nontemporal", [&](int k) { shape4_kernel<<<B, 128>>>(r[k], flow[k], cls[k], reg[k], mask[k], B); });
    run("shape SPB=1 nontemporal", [&](int k) { shape_kernel<1, true><<<B, 256>>>(r[k], flow[k], cls[k], reg[k], mask[k], B); });
    run("shape SPB=2 nontemporal", [&](int k) { shape_kernel<2, true><<<B / 2, 256>>>(r[k], flow[k], cls[k], reg[k], mask[k], B); });
    // flat: read 4N*B from r, write 28N*B into flow+reg+cls region (use flow/reg/cls as one? separate allocs: write cls+flow+reg sizes)
    for (int g : {1024, 2048, 4096, 8192}) {
        char nm[64]; snprintf(nm, sizeof nm, "flat grid=%d (r 4N, w 24N)", g);
        run(nm, [&](int k) {
            flat_kernel<<<g, 256>>>(reinterpret_cast<const float4 *>(r[k]), reinterpret_cast<float4 *>(flow[k]),
                                    (long long)B * N / 4, (long long)B * N * 2 / 4);
            flat_kernel<<<g, 256>>>(reinterpret_cast<const float4 *>(r[k]), reinterpret_cast<float4 *>(reg[k]),
                                    0, (long long)B * N * 2 / 4);
            flat_kernel<<<g, 256>>>(reinterpret_cast<const float4 *>(r[k]), reinterpret_cast<float4 *>(cls[k]),
                                    0, (long long)B * N * 2 / 4);
        });
    }
    // single large pure write and pure read, for reference
    {
        float4 *big; const size_t nb = (size_t)1 << 30; CK(hipMalloc(&big, nb)); CK(hipMemset(big, 0, nb));
        for (int i = 0; i < 3; ++i) flat_kernel<<<8192, 256>>>(big, big, 0, nb / 16);
        CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) flat_kernel<<<8192, 256>>>(big, big, 0, nb / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pure write 1 GiB: %.0f GB/s\n", nb / (ms / 10 * 1e-3) / 1e9);
        float4 *sink; CK(hipMalloc(&sink, (size_t)8192 * 256 * 16));  // one float4 per thread
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) flat_kernel<<<8192, 256>>>(big, sink, nb / 16, 8192 * 256);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pure read 1 GiB: %.0f GB/s\n", nb / (ms / 10 * 1e-3) / 1e9);
    }
    return 0;
}
