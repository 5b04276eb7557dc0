// Issue cadence of a lone wave / several waves per SIMD on gfx950: dependent vs independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters) {
    double a = out[threadIdx.x], b = a + 1.0, c = a + 2.0, d = a + 3.0;
    double m = 1.0000001, n = 0.5;
    int ia = (int)a, ib = ia + 1, ic = ia + 2, id = ia + 3;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // one dependent f64 fma chain
            REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(n));)
        } else if (MODE == 1) {  // two independent chains interleaved
            REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(m), "v"(n));)
        } else if (MODE == 2) {  // four independent chains
            REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n));)
        } else if (MODE == 3) {  // dependent f64 add chain
            REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(n));)
        } else if (MODE == 4) {  // dependent f64 mul chain
            REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(m));)
        } else if (MODE == 5) {  // dependent 32-bit cndmask chain
            REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ib) : );)
        } else if (MODE == 6) {  // independent 32-bit cndmask
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia));)
        } else if (MODE == 7) {  // dependent f32 fma chain
            float fa = (float)a;
            REP64(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(fa));)
            a = fa;
        } else if (MODE == 8) {  // f64 fma then dependent cndmask pairs (mix)
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(m), "+v"(n), "+v"(ia), "+v"(ib));)
        } else if (MODE == 9) {  // dpp mov chain
            REP64(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(ia));)
        } else if (MODE == 10) {  // 2 indep fma chains with SGPR addend
            REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(m), "s"(n));)
        } else if (MODE == 11) {  // v_cmp f64 + cndmask dependent
            REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f64 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(a), "+v"(b), "+v"(ia), "+v"(ib) : : "vcc");)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + ia + ib + ic + id;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int threads) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMemset(out, 0, sizeof(double) * blocks * threads);
    int nw = blocks * threads / 64;
    hipMalloc(&cyc, sizeof(long long) * nw);
    const int iters = 200;
    k<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= nw;
    // readcyclecounter = s_memtime at 100 MHz constant clock? report both
    printf("%-34s blocks %5d thr %4d: counter ticks/instr %.3f   wall ns/instr(per wave) %.3f\n", name, blocks, threads,
           mean / (iters * 64.0), ms * 1e6 / (iters * 64.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    // one wave on the chip; then 1/2/4 waves per SIMD on one CU (256 threads = 4 waves = 1 per SIMD)
    for (int cfg = 0; cfg < 3; ++cfg) {
        int blocks = 1, threads = cfg == 0 ? 64 : (cfg == 1 ? 256 : 512);
        printf("--- %d threads in one workgroup (%s)\n", threads, cfg == 0 ? "lone wave" : cfg == 1 ? "1 wave per SIMD" : "2 waves per SIMD");
        run<0>("dep fma_f64", blocks, threads);
        run<1>("2 indep fma_f64", blocks, threads);
        run<2>("4 indep fma_f64", blocks, threads);
        run<10>("2 indep fma_f64 sgpr addend", blocks, threads);
        run<3>("dep add_f64", blocks, threads);
        run<4>("dep mul_f64", blocks, threads);
        run<5>("dep cndmask_b32", blocks, threads);
        run<6>("4 indep cndmask_b32", blocks, threads);
        run<7>("dep fma_f32", blocks, threads);
        run<8>("fma_f64 + 3 cndmask", blocks, threads);
        run<9>("dep mov_dpp", blocks, threads);
        run<11>("cmp_f64+cndmask", blocks, threads);
    }
    return 0;
}
