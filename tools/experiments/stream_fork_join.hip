// What does a per-iteration fork / join between two HIP streams cost?  (Adam loop: the adjoint + update of channel c on stream A, the
// forward boxes of channel c on stream B as soon as channel c is updated, the warp after both.)  Kernels that spin for a given time stand
// in for the real ones; reported: us per iteration of the serial chain on one stream, of the two-stream schedule with events, and of the
// same schedule captured into a hipGraph.
//   hipcc --offload-arch=gfx950 -O3 stream_fork_join.hip -o /tmp/sfj && /tmp/sfj
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_spin(float* p, long long clocks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < clocks) {}
    if (clocks < 0) p[0] = 1.f;
}
static void spin(hipStream_t s, float* p, double us, int blocks) { hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, s, p, (long long)(us * 100.0)); }   // wall_clock64: 100 MHz

int main() {
    float* p; hipMalloc(&p, 4);
    hipStream_t A, B; hipStreamCreateWithFlags(&A, hipStreamNonBlocking); hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    hipEvent_t t0, t1, e[8]; hipEventCreate(&t0); hipEventCreate(&t1);
    for (auto& x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
    const int N = 200;
    const double adj = 5.6, fwd = 5.0, warp = 21.8;          // per channel: adjoint + update 16.7 / 3, forward boxes 15 / 3
    auto serial = [&](hipStream_t s) {
        for (int i = 0; i < N; ++i) { spin(s, p, 3 * adj, 1024); spin(s, p, 3 * fwd, 252); spin(s, p, warp, 1024); }
    };
    auto forked = [&](hipStream_t a, hipStream_t b) {
        for (int i = 0; i < N; ++i) {
            for (int c = 0; c < 3; ++c) {
                spin(a, p, adj, 340);
                hipEventRecord(e[c], a);
                hipStreamWaitEvent(b, e[c], 0);
                spin(b, p, fwd, 84);
            }
            hipEventRecord(e[3], b);
            hipStreamWaitEvent(a, e[3], 0);
            spin(a, p, warp, 1024);
            hipEventRecord(e[4], a);
            hipStreamWaitEvent(b, e[4], 0);       // the next iteration's forward boxes must not overtake this warp (they overwrite U)
        }
    };
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(t0, A); serial(A); hipEventRecord(t1, A); hipEventSynchronize(t1); hipEventElapsedTime(&ms, t0, t1);
        printf("one stream, 3 launches per iteration: %.1f us per iteration (kernels alone: %.1f)\n", ms * 1e3 / N, 3 * adj + 3 * fwd + warp);
        hipEventRecord(t0, A); forked(A, B); hipEventRecord(t1, A); hipEventSynchronize(t1); hipStreamSynchronize(B); hipEventElapsedTime(&ms, t0, t1);
        printf("two streams, per-channel fork + join with events, 7 launches per iteration: %.1f us per iteration (critical path: %.1f)\n", ms * 1e3 / N, 3 * adj + fwd + warp);
    }
    // the same schedule as a graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(A, hipStreamCaptureModeGlobal);
    hipEventRecord(e[5], A); hipStreamWaitEvent(B, e[5], 0);
    forked(A, B);
    hipEventRecord(e[6], B); hipStreamWaitEvent(A, e[6], 0);
    hipStreamEndCapture(A, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(t0, A); hipGraphLaunch(ge, A); hipEventRecord(t1, A); hipEventSynchronize(t1); hipEventElapsedTime(&ms, t0, t1);
        printf("the two-stream schedule captured into a hipGraph: %.1f us per iteration\n", ms * 1e3 / N);
    }
    return 0;
}
