// Does an event pair recorded INSIDE a captured stream give usable timings when the graph is replayed?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float *p, int n) { float x = p[threadIdx.x]; for (int i = 0; i < n; i++) x = x * 1.0001f + 0.5f; p[threadIdx.x] = x; }
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); return 1; } } while (0)
int main() {
    float *d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 1000);
    CK(hipEventRecord(a, st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 200000);
    CK(hipEventRecord(b, st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 1000);
    hipGraph_t g; CK(hipStreamEndCapture(st, &g));
    hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; i++) {
        CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
        float ms = -1; hipError_t r = hipEventElapsedTime(&ms, a, b);
        printf("launch %d: elapsed %s %.3f ms\n", i, hipGetErrorString(r), ms);
    }
    return 0;
}
