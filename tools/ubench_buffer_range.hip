// Check: raw buffer stores with a per-row range drop the lanes beyond it (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(double* out, unsigned* out32, const int* n_in) {
    const int lane = threadIdx.x & 63;
    const int np = __builtin_amdgcn_readfirstlane(n_in[blockIdx.x]);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.x * 64, 0, np * 8, 0x00020000);
    __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(out32 + (size_t)blockIdx.x * 64, 0, np * 4, 0x00020000);
    double v = 1.5 * lane + blockIdx.x;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, lane * 8, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)(lane + 1), r2, lane * 4, 0, 0);
}
int main() {
    const int nb = 70;
    std::vector<int> n(nb);
    for (int b = 0; b < nb; ++b) n[b] = b % 65;   // 0..64 lanes in range
    double* out; unsigned* out32; int* dn;
    hipMalloc(&out, sizeof(double) * nb * 64); hipMalloc(&out32, 4 * nb * 64); hipMalloc(&dn, 4 * nb);
    hipMemset(out, 0xff, sizeof(double) * nb * 64); hipMemset(out32, 0xff, 4 * nb * 64);
    hipMemcpy(dn, n.data(), 4 * nb, hipMemcpyHostToDevice);
    k<<<nb, 64>>>(out, out32, dn);
    std::vector<double> h(nb * 64); std::vector<unsigned> h32(nb * 64);
    hipMemcpy(h.data(), out, sizeof(double) * nb * 64, hipMemcpyDeviceToHost);
    hipMemcpy(h32.data(), out32, 4 * nb * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < nb; ++b)
        for (int l = 0; l < 64; ++l) {
            const bool in = l < n[b];
            unsigned long long bits; std::memcpy(&bits, &h[b * 64 + l], 8);
            if (in ? h[b * 64 + l] != 1.5 * l + b : bits != ~0ull) ++bad;
            if (in ? h32[b * 64 + l] != (unsigned)(l + 1) : h32[b * 64 + l] != ~0u) ++bad;
        }
    printf("raw buffer store range check: %d mismatches (%s)\n", bad, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
