// Wave launch rate of the chip: how long do N short-lived workgroups take when they do (next to) nothing?
//   hipcc --offload-arch=gfx950 -O3 -o ubench_dispatch tools/ubench_dispatch.hip && ./ubench_dispatch
// Variants: threads per workgroup, LDS per workgroup, a register floor, and a fixed life per wave (s_sleep loops) --
// k_assoc_group launches 25 000 workgroups of 4 waves (100 000 waves, 14 KB of LDS each) for ~9 us of life per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDS_BYTES, int SLEEPS>
__global__ void k_empty(int* out, int n) {
    extern __shared__ char dyn[];
    __shared__ char pad[LDS_BYTES > 0 ? LDS_BYTES : 1];
    if (LDS_BYTES > 0) pad[threadIdx.x] = (char)threadIdx.x;
    for (int i = 0; i < SLEEPS; ++i) __builtin_amdgcn_s_sleep(127);   // ~127 x 64 cycles each
    if (out && blockIdx.x == 0x7fffffff) out[0] = pad[0] + n;
}

template <class F>
static float time_ms(F f, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    int* out; hipMalloc(&out, 64);
    const int waves = 100000;
    struct V { const char* name; int threads; };
    for (int threads : {64, 256, 1024}) {
        const int blocks = waves * 64 / threads;
        float t0 = time_ms([&] { k_empty<0, 0><<<blocks, threads>>>(out, 1); });
        float t1 = time_ms([&] { k_empty<14336, 0><<<blocks, threads>>>(out, 1); });
        float t2 = time_ms([&] { k_empty<14336, 1><<<blocks, threads>>>(out, 1); });     // ~3.4 us of life
        float t3 = time_ms([&] { k_empty<14336, 3><<<blocks, threads>>>(out, 1); });     // ~10 us of life
        printf("%6d waves as %6d workgroups of %4d threads: empty %.1f us | 14 KB LDS %.1f us | + 1 sleep(127) %.1f us | + 3 sleeps %.1f us\n",
               waves, blocks, threads, t0 * 1e3, t1 * 1e3, t2 * 1e3, t3 * 1e3);
    }
    return 0;
}
