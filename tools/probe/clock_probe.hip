// Clock / VALU issue-rate probe (diagnostic, not part of the library): dependent and independent v_fma_f32 chains at 1, 2 and 4
// waves per SIMD, timed with HIP events, for short (~50 us) and long (~5 ms) launches; s_memtime deltas beside them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int ILP>
__global__ __launch_bounds__(256) void spin(float* out, int n, unsigned long long* ticks)
{
    float a[ILP];
    for (int k = 0; k < ILP; ++k) a[k] = (float)threadIdx.x + k;
    const float b = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / ILP; ++r)
#pragma unroll
            for (int k = 0; k < ILP; ++k) a[k] = __builtin_fmaf(a[k], b, c);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int k = 0; k < ILP; ++k) s += a[k];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main()
{
    float* out; unsigned long long* ticks;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&ticks, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    for (int n : {2000, 200000}) {
        for (int wps : {1, 2, 4}) {
            for (int ilp : {1, 4}) {
                const int blocks = p.multiProcessorCount * wps;
                float best = 1e30f; unsigned long long tk = 0;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(e0));
                    if (ilp == 1) spin<1><<<blocks, 256>>>(out, n, ticks); else spin<4><<<blocks, 256>>>(out, n, ticks);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) { best = ms; CK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost)); }
                }
                const double insts = (double)n * 16.0;  // per wave
                printf("n=%7d waves/SIMD=%d ilp=%d: %.4f ms  -> %.2f ns per wave-instr, x waves/SIMD: %.3f ns per SIMD issue slot; s_memtime ticks %llu (%.1f MHz)\n", n, wps, ilp, best,
                       best * 1e6 / insts, best * 1e6 / insts / wps, tk, tk / (best * 1e3));
            }
        }
    }
    return 0;
}
