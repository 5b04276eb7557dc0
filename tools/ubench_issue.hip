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
        } else if (MODE == 12) {  // four independent f32 fma chains
            float fa = (float)a, fb = (float)b, fc = (float)c, fd = (float)d;
            REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
            a = fa; b = fb; c = fc; d = fd;
        } else if (MODE == 13) {  // four independent packed-f32 fma chains (two f32 per instruction and lane)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (MODE == 14) {  // four independent dpp movs
            REP16(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));)
        } else if (MODE == 15) {  // four independent f64 adds
            REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(n));)
        } else if (MODE == 16) {  // f64 compares into SGPR pairs, independent
            REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %3, %0" : : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");)
        } else if (MODE == 17) {  // the association's mix: 2 f64 : 1 dpp : 4 32-bit, independent
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %5\n v_mul_f64 %1, %1, %8\n v_cndmask_b32 %5, %5, %6, vcc\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_and_b32 %7, %7, %4\n v_lshlrev_b32 %4, 1, %4\n v_add_f64 %2, %2, %9"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(m), "v"(n));)
        } else if (MODE == 18) {  // scalar ALU beside vector: 1 s_add : 1 v_fma_f64
            int sa = iters;
            REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n s_add_u32 %6, %6, 1\n v_fma_f64 %1, %1, %4, %5\n s_add_u32 %6, %6, 1\n v_fma_f64 %2, %2, %4, %5\n s_add_u32 %6, %6, 1\n v_fma_f64 %3, %3, %4, %5\n s_add_u32 %6, %6, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n), "s"(sa) : "scc");)
        } else if (MODE == 19) {  // scalar ALU only: four independent s_add chains
            int sa = iters, sb2 = iters + 1, sc = iters + 2, sd = iters + 3;
            REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(sa), "+s"(sb2), "+s"(sc), "+s"(sd) : : "scc");)
            ia += sa + sb2 + sc + sd;
        } else if (MODE == 20) {  // scalar-heavy mix: 3 scalar : 1 vector (s_and_b64 / s_cmp / s_add : v_fma_f64)
            int sa = iters;
            REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n s_add_u32 %4, %4, 1\n s_and_b64 vcc, vcc, exec\n s_cmp_lg_u32 %4, 0\n v_fma_f64 %1, %1, %2, %3\n s_add_u32 %4, %4, 1\n s_and_b64 vcc, vcc, exec\n s_cmp_lg_u32 %4, 0" : "+v"(a), "+v"(b) : "v"(m), "v"(n), "s"(sa) : "scc", "vcc");)
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
    // per SIMD: waves on a SIMD = (waves per workgroup / 4) x workgroups per CU (256 CUs)
    const double wps = (threads / 64) / 4.0 * (blocks <= 256 ? 1.0 : blocks / 256.0);
    const int per_iter = MODE == 17 || MODE == 18 || MODE == 20 ? 128 : 64;
    printf("%-34s blocks %5d thr %4d: counter ticks/instr %.3f   wall ns/instr(per wave) %.3f   SIMD ns/instr %.3f (= %.2f cyc at 2.4 GHz)\n", name, blocks, threads,
           mean / (iters * (double)per_iter), ms * 1e6 / (iters * (double)per_iter), ms * 1e6 / (iters * (double)per_iter) / (wps < 1 ? 1 : wps),
           ms * 1e6 / (iters * (double)per_iter) / (wps < 1 ? 1 : wps) * 2.4);
    hipFree(out); hipFree(cyc);
}

int main() {
    // one wave on the chip; then 1/2/4 waves per SIMD on one CU (256 threads = 4 waves = 1 per SIMD)
    for (int cfg = 0; cfg < 5; ++cfg) {
        int blocks = cfg < 3 ? 1 : (cfg == 3 ? 256 * 4 : 256 * 8), threads = cfg == 0 ? 64 : (cfg == 2 ? 512 : 256);
        printf("--- %d workgroup(s) of %d threads (%s)\n", blocks, threads, cfg == 0 ? "lone wave" : cfg == 1 ? "1 wave per SIMD" : cfg == 2 ? "2 waves per SIMD" :
               cfg == 3 ? "whole chip, 4 waves per SIMD" : "whole chip, 8 waves per SIMD");
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
        run<12>("4 indep fma_f32", blocks, threads);
        run<13>("4 indep pk_fma_f32", blocks, threads);
        run<14>("4 indep mov_dpp", blocks, threads);
        run<15>("4 indep add_f64", blocks, threads);
        run<16>("4 indep cmp_f64", blocks, threads);
        run<17>("mix 3 f64 : 1 dpp : 4 int (x16)", blocks, threads);
        run<18>("v_fma_f64 + s_add pairs", blocks, threads);
        run<19>("4 indep s_add_u32 (scalar only)", blocks, threads);
        run<20>("1 v_fma_f64 : 3 scalar", blocks, threads);
    }
    return 0;
}
