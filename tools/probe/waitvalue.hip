// probe: hipStreamWaitValue64 on plain device memory vs signal memory; cost of device atomics on signal memory
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_add(unsigned long long* p, int n, unsigned long long* out)
{
    unsigned long long acc = 0;
    for (int i = 0; i < n; ++i) acc += atomicAdd(p, 1ull);
    if (out) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_set(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
__global__ void k_mark(unsigned long long* p) { *p = 1; }
int main()
{
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CanUseStreamWaitValue %d\n", can);
    unsigned long long *plain, *sig = nullptr, *out, *mark;
    hipMalloc(&plain, 64); hipMemset(plain, 0, 64);
    hipMalloc(&out, 1 << 20); hipMalloc(&mark, 8); hipMemset(mark, 0, 8);
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("signal alloc: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) hipMemset(sig, 0, 8);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    for (int which = 0; which < 2; ++which) {
        unsigned long long* p = which ? sig : plain;
        if (!p) continue;
        // latency of returning atomics: one lane, 1000 in a row
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_add, dim3(1), dim3(1), 0, a, p, 1000, out);
        hipStreamSynchronize(a);
        auto t1 = std::chrono::steady_clock::now();
        printf("%s: 1000 dependent returning atomics %.1f us\n", which ? "signal" : "plain", std::chrono::duration<double, std::micro>(t1 - t0).count());
        // wait-value: stream b waits until *p >= 5000, then marks; stream a adds 5000 later
        hipMemsetAsync(mark, 0, 8, b); hipStreamSynchronize(b);
        e = hipStreamWaitValue64(b, p, 5000ull, hipStreamWaitValueGte, ~0ull);
        printf("%s: hipStreamWaitValue64 -> %s\n", which ? "signal" : "plain", hipGetErrorString(e));
        if (e != hipSuccess) { (void)hipGetLastError(); continue; }
        hipLaunchKernelGGL(k_mark, dim3(1), dim3(1), 0, b, mark);
        unsigned long long m = 7;
        hipMemcpy(&m, mark, 8, hipMemcpyDeviceToHost);
        printf("  mark before the add: %llu (0 expected)\n", m);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, a, p, 5000ull);
        hipStreamSynchronize(a);
        auto t2 = std::chrono::steady_clock::now();
        hipStreamSynchronize(b);
        auto t3 = std::chrono::steady_clock::now();
        hipMemcpy(&m, mark, 8, hipMemcpyDeviceToHost);
        printf("  mark after the add: %llu (1 expected), released %.1f us after the adding kernel was done\n", m, std::chrono::duration<double, std::micro>(t3 - t2).count());
    }
    return 0;
}
