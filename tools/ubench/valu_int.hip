// Micro-benchmark: issue cost of the integer instructions the field arithmetic is made of, on gfx950.
// For each op: ITER iterations of UNROLL instructions on NCH independent chains, W waves per SIMD.
// Prints ns and shader cycles (s_memtime) per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2000;

template <int OP, int NCH>
__global__ void __launch_bounds__(64) k(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
    uint32_t a[NCH], b[NCH];
    uint64_t c[NCH];
    double d[NCH];
    for (int i = 0; i < NCH; i++) { a[i] = seed * (i + 3) + threadIdx.x; b[i] = seed * 7 + i * 11 + threadIdx.x; c[i] = a[i]; d[i] = 1.0 + a[i] * 1e-9; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                if (OP == 0) c[i] = (uint64_t)a[i] * (uint32_t)c[i] + c[i];          // v_mad_u64_u32, dependent through c
                if (OP == 1) a[i] = a[i] * b[i] + 1;                                  // v_mul_lo_u32 (+add)
                if (OP == 2) a[i] = __umulhi(a[i], b[i]) + a[i];                      // v_mul_hi_u32
                if (OP == 3) a[i] = a[i] + b[i];                                      // v_add_u32
                if (OP == 4) c[i] = c[i] + (c[i] << 1);                               // v_lshl_add_u64
                if (OP == 5) d[i] = __builtin_fma(d[i], 1.0000001, 0.5);              // v_fma_f64
                if (OP == 6) a[i] = ((a[i] & 0xffffff) * (b[i] & 0xffffff)) + a[i];      // v_mad_u32_u24
                if (OP == 7) { asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(a[i]), "v"(b[i]) : "vcc"); } // add/addc pair
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int i = 0; i < NCH; i++) acc += a[i] + b[i] + (uint32_t)c[i] + (uint32_t)(c[i] >> 32) + (uint32_t)d[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int NCH>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd;      // one wave per block
    uint32_t* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, blocks * 64 * 4)); CHECK(hipMalloc(&cyc, blocks * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP, NCH>), dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP, NCH>), dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double ninstr = (double)ITER * 8 * NCH * (OP == 7 ? 2 : 1);
    // per SIMD: waves_per_simd waves each issue ninstr instructions in `ms`
    printf("%-14s chains=%d waves/SIMD=%d : %.2f ms, %.2f ns per wave-instr per SIMD, %.1f memtime-ticks per instr (one wave's view), %.2f Ginstr/s chip\n",
           name, NCH, waves_per_simd, ms, ms * 1e6 / (ninstr * waves_per_simd), avg / ninstr, ninstr * blocks / ms / 1e6);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0, 1>("mad_u64_u32", w); run<0, 4>("mad_u64_u32", w);
        run<1, 4>("mul_lo_u32", w); run<2, 4>("mul_hi_u32", w);
        run<3, 4>("add_u32", w); run<4, 4>("lshl_add_u64", w); run<5, 4>("fma_f64", w);
        run<6, 4>("mad_u32_u24", w); run<7, 4>("add_co/addc", w);
    }
    return 0;
}
