// Ablation of the headline launch (pof_scan_preprocess_chained, float32 outputs, B = 4096 x 450 points)
// through the library's own C ABI: which outputs / how many detections cost what.  Links libpof_hip.so.
// Build: hipcc --offload-arch=gfx950 -O3 -I include -o headline_abl tools/ubench/headline_abl.hip \
//        -L planar_optical_flow_amd/lib -lpof_hip -Wl,-rpath,$PWD/planar_optical_flow_amd/lib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>

#include "pof_abi.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Slot {
    float *scans; double *o0, *o1; int32_t *offs; double *rphi; uint8_t *cls; int D;
    float *flow, *reg, *mask; int64_t *tcls; void *ws; size_t ws_bytes;
};

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096, N = 450, RING = 8, ITERS = 400;
    // argv[2] = number of streams: batch i runs on stream i % NS and evaluates the params of batch i + NS, so
    // that consecutive launches (independent batches) may overlap their ramp and tail
    const int NS = argc > 2 ? atoi(argv[2]) : 1;
    const int GRAPH = argc > 3 ? atoi(argv[3]) : 0;     // > 0: replay a captured graph of GRAPH trips round the ring
    hipStream_t streams[4];
    for (int i = 0; i < NS; ++i) CK(hipStreamCreate(&streams[i]));
    double *tab; CK(hipMalloc(&tab, 3 * N * sizeof(double)));
    if (pof_laser_phi(0.5 * M_PI / 180.0, N, tab, nullptr)) return 1;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> ur(0.5f, 25.f);
    std::uniform_real_distribution<double> uo(-0.05, 0.05), ud(1.0, 10.0), up(-1.9, 1.9);
    const double ar[3] = {0.6, 0.4, 0.35}, dr[3] = {2.5, 2.0, 2.0};
    const int32_t lb[3] = {1, 2, 3};
    for (int ndet : {3, 0, 8, -1}) {           // detections per sample; -1: no detection arguments at all
        std::vector<Slot> ring(RING);
        for (auto &s : ring) {
            std::vector<float> h((size_t)B * 2 * N);
            for (auto &v : h) v = ur(rng);
            std::vector<double> a(3 * B), b(3 * B);
            for (auto &v : a) v = uo(rng);
            for (auto &v : b) v = uo(rng);
            const int per = ndet < 0 ? 0 : ndet;
            std::vector<int32_t> offs(B + 1);
            for (int i = 0; i <= B; ++i) offs[i] = i * per;
            s.D = B * per;
            std::vector<double> rp(2 * (size_t)std::max(s.D, 1));
            for (size_t i = 0; i < rp.size(); i += 2) { rp[i] = ud(rng); rp[i + 1] = up(rng); }
            std::vector<uint8_t> cl(std::max(s.D, 1), 2);
            CK(hipMalloc(&s.scans, h.size() * 4)); CK(hipMemcpy(s.scans, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            CK(hipMalloc(&s.o0, a.size() * 8)); CK(hipMemcpy(s.o0, a.data(), a.size() * 8, hipMemcpyHostToDevice));
            CK(hipMalloc(&s.o1, b.size() * 8)); CK(hipMemcpy(s.o1, b.data(), b.size() * 8, hipMemcpyHostToDevice));
            CK(hipMalloc(&s.offs, offs.size() * 4)); CK(hipMemcpy(s.offs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice));
            CK(hipMalloc(&s.rphi, rp.size() * 8)); CK(hipMemcpy(s.rphi, rp.data(), rp.size() * 8, hipMemcpyHostToDevice));
            CK(hipMalloc(&s.cls, cl.size())); CK(hipMemcpy(s.cls, cl.data(), cl.size(), hipMemcpyHostToDevice));
            CK(hipMalloc(&s.flow, (size_t)B * N * 8)); CK(hipMalloc(&s.reg, (size_t)B * N * 8));
            CK(hipMalloc(&s.tcls, (size_t)B * N * 8)); CK(hipMalloc(&s.mask, (size_t)B * N * 4));
            s.ws_bytes = pof_scan_preprocess_workspace_bytes(B, s.D);
            CK(hipMalloc(&s.ws, s.ws_bytes));
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto run = [&](const char *name, bool flow, bool assoc, bool mask, double bytes_per_pt) {
            auto step = [&](int i) {
                Slot &s = ring[i % RING], &n = ring[(i + NS) % RING];
                hipStream_t st = NS > 1 ? streams[i % NS] : nullptr;
                const bool dets = ndet >= 0 && (assoc || mask);
                pof_scan_inputs nx = {};
                nx.odom0 = n.o0; nx.odom1 = n.o1; nx.B = B; nx.want_flow = flow;
                if (dets) { nx.det_offsets = n.offs; nx.det_rphi = n.rphi; nx.det_cls = n.cls; nx.D = n.D; }
                for (int k = 0; k < 3; ++k) { nx.assoc_radius[k] = ar[k]; nx.dyn_radius[k] = dr[k]; nx.labels[k] = lb[k]; }
                nx.workspace = n.ws; nx.workspace_bytes = n.ws_bytes;
                int rc = pof_scan_preprocess_chained(s.scans + N, 2 * N, B, N, tab, s.o0, s.o1, 0, 1, 0, nullptr,
                                                     flow ? s.flow : nullptr, dets ? s.offs : nullptr, s.rphi, s.cls, s.D, ar, lb, dr,
                                                     nullptr, (dets && assoc) ? s.tcls : nullptr, (dets && assoc) ? s.reg : nullptr, nullptr, nullptr,
                                                     mask ? s.mask : nullptr, s.ws, s.ws_bytes, &nx, st);
                if (rc) { printf("rc %d\n", rc); exit(1); }
            };
            // prime: params of the first NS slots
            for (int k = 0; k < NS; ++k) {
                Slot &s = ring[k];
                pof_scan_preprocess_phase(s.scans + N, 2 * N, B, N, tab, s.o0, s.o1, 0, 1, 0, nullptr, s.flow,
                                          ndet >= 0 ? s.offs : nullptr, s.rphi, s.cls, s.D, ar, lb, dr, nullptr, nullptr, nullptr,
                                          nullptr, nullptr, nullptr, s.ws, s.ws_bytes, 1, nullptr);
            }
            for (int i = 0; i < 2 * RING; ++i) step(i);
            CK(hipDeviceSynchronize());
            float ms;
            if (GRAPH && NS > 1) {
                // one trip round the ring captured as a graph with NS parallel branches (fork / join by events)
                hipGraph_t g; hipGraphExec_t ge;
                hipEvent_t fork, join[4];
                CK(hipEventCreate(&fork));
                for (int k = 0; k < NS; ++k) CK(hipEventCreate(&join[k]));
                CK(hipStreamBeginCapture(streams[0], hipStreamCaptureModeGlobal));
                CK(hipEventRecord(fork, streams[0]));
                for (int k = 1; k < NS; ++k) CK(hipStreamWaitEvent(streams[k], fork, 0));
                for (int i = 0; i < GRAPH * RING; ++i) step(i);
                for (int k = 1; k < NS; ++k) { CK(hipEventRecord(join[k], streams[k])); CK(hipStreamWaitEvent(streams[0], join[k], 0)); }
                CK(hipStreamEndCapture(streams[0], &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, streams[0]));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, streams[0]));
                const int reps = ITERS / (GRAPH * RING);
                for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, streams[0]));
                CK(hipEventRecord(e1, streams[0])); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms = ms * ITERS / (reps * GRAPH * RING);
            } else {
                CK(hipEventRecord(e0));
                for (int i = 0; i < ITERS; ++i) step(i);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double us = ms / ITERS * 1e3;
            printf("dets/sample %2d  %-34s %7.2f us  %6.0f GB/s (%.0f B/pt)\n", ndet, name, us,
                   bytes_per_pt * B * N / (us * 1e-6) / 1e9, bytes_per_pt);
            fflush(stdout);
        };
        run("flow+cls+reg+mask (headline)", true, true, true, 32);
        run("flow only", true, false, false, 12);
        run("flow+mask", true, false, true, 16);
        run("cls+reg+mask (no flow)", false, true, true, 24);
        for (auto &s : ring) {
            hipFree(s.scans); hipFree(s.o0); hipFree(s.o1); hipFree(s.offs); hipFree(s.rphi); hipFree(s.cls);
            hipFree(s.flow); hipFree(s.reg); hipFree(s.tcls); hipFree(s.mask); hipFree(s.ws);
        }
    }
    return 0;
}
