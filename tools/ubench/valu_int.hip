// Micro-benchmark: ISSUE cost of the integer instructions the field arithmetic is made of, on gfx950, at 1 / 2 / 4 / 8
// wavefronts per SIMD.  Every instruction is an `asm volatile` whose operands depend on values loaded at run time, so
// the compiler can neither fold nor hoist it (round 1's version let the cheap opcodes be folded away).  NCH independent
// dependency chains per wave (8: throughput; 1: dependent-issue latency).
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_int.hip -o tools/ubench/valu_int && tools/ubench/valu_int > profiles/rNN_ubench_valu_int.csv
//
// Output: CSV  op,chains,waves_per_simd,ns_per_wave_instr_per_simd,cycles_per_wave_instr_per_simd,clock_ghz
// (cycles = ns x clock_ghz, the shader clock as s_memtime ticks of one wavefront / kernel wall time: only the
// 1-wave-per-SIMD rows give a usable clock -- with more waves than SIMDs the waves of one launch do not all live for
// the whole kernel; tools/valu_model.py works from the ns column).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 1500;
constexpr int UNR = 8;

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, ADD_U32, SUB_U32, AND_B32, OR_B32, LSHRREV, LSHLREV, CNDMASK, MOV, ADD3, LSHL_ADD, LSHL_OR,
          AND_OR, BFE, ALIGNBIT, ADDCO_ADDC, LSHL_ADD_U64, BPERMUTE, MAD_MIX, N_OPS };
static const char* NAMES[N_OPS] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32",
                                   "v_lshrrev_b32", "v_lshlrev_b32", "v_cndmask_b32", "v_mov_b32", "v_add3_u32", "v_lshl_add_u32", "v_lshl_or_b32",
                                   "v_and_or_b32", "v_bfe_u32", "v_alignbit_b32", "v_add_co_u32+v_addc_co_u32", "v_lshl_add_u64", "ds_bpermute_b32",
                                   "mix:3xv_mad_u64_u32+1xv_and_b32"};

template <int OP, int NCH>
__global__ void __launch_bounds__(64) k(uint32_t* out, unsigned long long* cyc, const uint32_t* in) {
    uint32_t a[NCH], b[NCH];
    uint64_t c[NCH];
    for (int i = 0; i < NCH; i++) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * (i + NCH)] | 1u; c[i] = ((uint64_t)b[i] << 32) | a[i]; }
    const uint32_t sh = in[0] & 7u;                   // run-time shift amount
    const uint64_t mask = __ballot((in[threadIdx.x] >> 3) & 1u);   // run-time lane mask in an SGPR pair
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
                if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == OR_B32) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == LSHRREV) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(sh));
                if (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(sh));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(mask));
                if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 1, 28" : "+v"(a[i]));
                if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(a[i]) : "v"(b[i]));
                if (OP == ADDCO_ADDC) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(sh) : "vcc");
                if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(c[i]));
                if (OP == BPERMUTE) { asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(b[i])); }
                if (OP == MAD_MIX) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_and_b32 %1, %1, %2"
                                                : "+v"(c[i]), "+v"(a[i]) : "v"(b[i]) : "vcc");
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int i = 0; i < NCH; i++) acc += a[i] + b[i] + (uint32_t)c[i] + (uint32_t)(c[i] >> 32);
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static uint32_t* g_in;

template <int OP, int NCH>
void run(int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd;      // one wavefront per block, `waves_per_simd` per SIMD
    uint32_t* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, blocks * 64 * 4)); CHECK(hipMalloc(&cyc, blocks * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP, NCH>), dim3(blocks), dim3(64), 0, 0, out, cyc, g_in);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP, NCH>), dim3(blocks), dim3(64), 0, 0, out, cyc, g_in);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const int per = (OP == ADDCO_ADDC) ? 2 : (OP == MAD_MIX ? 4 : 1);
    const double ninstr = (double)ITER * UNR * NCH * per;
    const double ns = ms * 1e6 / (ninstr * waves_per_simd);
    const double ghz = avg / (ms * 1e6);              // s_memtime ticks per ns over one wave's lifetime ~ kernel time
    printf("%s,%d,%d,%.3f,%.2f,%.2f\n", NAMES[OP], NCH, waves_per_simd, ns, ns * ghz, ghz);
    fflush(stdout);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

template <int OP>
void sweep() {
    for (int w : {1, 2, 4, 8}) { run<OP, 8>(w); }
    run<OP, 1>(1);                                    // one dependent chain: issue-to-issue latency
}

int main() {
    std::vector<uint32_t> h(64 * 32);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345u) >> 4;
    CHECK(hipMalloc(&g_in, h.size() * 4));
    CHECK(hipMemcpy(g_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // warm up: the first kernels of a process run before the clock has ramped (round 2's first row read 4.2 ns for an
    // instruction that measures 2.5 ns once warm)
    {
        const int wb = 256 * 4 * 4;
        uint32_t* wout; unsigned long long* wcyc;
        CHECK(hipMalloc(&wout, wb * 64 * 4)); CHECK(hipMalloc(&wcyc, wb * 8));
        for (int r = 0; r < 40; r++) hipLaunchKernelGGL((k<MAD_MIX, 8>), dim3(wb), dim3(64), 0, 0, wout, wcyc, g_in);
        CHECK(hipDeviceSynchronize());
        CHECK(hipFree(wout)); CHECK(hipFree(wcyc));
    }
    printf("op,chains,waves_per_simd,ns_per_wave_instr_per_simd,cycles_per_wave_instr_per_simd,clock_ghz\n");
    sweep<MAD_U64_U32>(); sweep<MUL_LO_U32>(); sweep<MUL_HI_U32>(); sweep<MAD_U32_U24>(); sweep<ADD_U32>(); sweep<SUB_U32>();
    sweep<AND_B32>(); sweep<OR_B32>(); sweep<LSHRREV>(); sweep<LSHLREV>(); sweep<CNDMASK>(); sweep<MOV>(); sweep<ADD3>();
    sweep<LSHL_ADD>(); sweep<LSHL_OR>(); sweep<AND_OR>(); sweep<BFE>(); sweep<ALIGNBIT>(); sweep<ADDCO_ADDC>(); sweep<LSHL_ADD_U64>();
    sweep<BPERMUTE>(); sweep<MAD_MIX>();
    return 0;
}
