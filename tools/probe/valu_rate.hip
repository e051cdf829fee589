// What does one VALU instruction cost on gfx950, per SIMD, at 1 / 2 / 4 waves per SIMD -- and does a packed-f32
// instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two floats per lane) cost one issue slot or two?
// VERDICT round 3, item 2 proposes packing two voxels per lane into v_pk_* to halve the list kernel's VALU time; that
// pays only if a packed instruction runs at the rate of a scalar one.  This probe measures it instead of assuming it.
//
// One workgroup per CU (256 of them), 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD; every wave runs a loop of
// 64 instructions of ONE kind on 8 independent accumulators (or one, "dep": a dependent chain), kIters times, between two
// s_memtime stamps.  Printed: shader clocks per instruction per WAVE (the median wave), and per SIMD (= that divided by
// the waves per SIMD): the SIMD's issue interval for that instruction.
// hipcc -O3 --offload-arch=gfx950 tools/probe/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float vf2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 2000;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum Kind { ADD, PK_ADD, FMA, PK_FMA, MUL, PK_MUL, CNDMASK, CNDMASK_SGPR, PAIR_VCC, PAIR_SGPR, PAIR_SGPR_NOP, CMP, MAX, AND, ADD_U32, LSHL_ADD, CVT_I, SQRT, RCP, ADD_F64, CVT_F64, ADD_DEP, PK_ADD_DEP, MOV_DPP, PK_MOV };

template <int KIND>
__global__ __launch_bounds__(1024) void rate(unsigned long long* __restrict__ out, float seed) {
    float a[8];
    vf2 p[8];
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + (float)(threadIdx.x + i);
        p[i] = vf2{a[i], a[i] * 0.5f};
        d[i] = (double)a[i];
    }
    float d2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d2[i] = seed * (float)(i + 1) + (float)threadIdx.x * 0.5f;
    const float b = seed * 1.0001f + 1e-3f;
    const vf2 pb = {b, b};
    const double db = (double)b;
    unsigned long long mask = __builtin_amdgcn_read_exec() ^ (0x5555555555555555ull * (blockIdx.x & 1));
    unsigned long long masks[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_mov_b64 vcc, %0" ::"s"(mask) : "vcc");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        if constexpr (KIND == ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == PK_ADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            REP64(X)
#undef X
        } else if constexpr (KIND == PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            REP64(X)
#undef X
        } else if constexpr (KIND == PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
            REP64(X)
#undef X
        } else if constexpr (KIND == CNDMASK) {
            // ONE asm statement per 8 selects: a statement that names vcc as clobbered makes the compiler pad the next
            // reader with s_nop (VALU-writes-VCC hazard), which is not what is being measured
#define X8 asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n" \
                        "v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"  \
                        "v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"                                    \
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])   \
                        : "v"(b));
            X8 X8 X8 X8 X8 X8 X8 X8
#undef X8
        } else if constexpr (KIND == CNDMASK_SGPR) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask));
            REP64(X)
#undef X
        } else if constexpr (KIND == PAIR_VCC) {
            // the pattern the compiler emits for `x = c < b ? y : x`: compare into vcc, select from vcc (32 pairs = 64 instr)
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(d2[i]), "v"(b) : "vcc");
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == PAIR_SGPR) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 %1, %2, %3\n v_cndmask_b32_e64 %0, %0, %3, %1" : "+v"(a[i]), "=&s"(masks[i]) : "v"(d2[i]), "v"(b));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == PAIR_SGPR_NOP) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 %1, %2, %3\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %3, %1" : "+v"(a[i]), "=&s"(masks[i]) : "v"(d2[i]), "v"(b));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == CMP) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(masks[i]) : "v"(a[i]), "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == MAX) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == AND) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == LSHL_ADD) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == CVT_I) {
#define X(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            REP64(X)
#undef X
        } else if constexpr (KIND == SQRT) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            REP64(X)
#undef X
        } else if constexpr (KIND == RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP64(X)
#undef X
        } else if constexpr (KIND == ADD_F64) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
            REP64(X)
#undef X
        } else if constexpr (KIND == CVT_F64) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            REP64(X)
#undef X
        } else if constexpr (KIND == ADD_DEP) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == PK_ADD_DEP) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(pb));
            REP64(X)
#undef X
        } else if constexpr (KIND == MOV_DPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            REP64(X)
#undef X
        } else if constexpr (KIND == PK_MOV) {
#define X(i) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(p[i]) : "v"(pb));
            REP64(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    for (int i = 0; i < 8; ++i) s += (float)masks[i];
    if (s == 12345.678f) out[0] = 1;  // keep everything alive
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, unsigned long long* dev) {
    printf("%-14s", name);
    for (int threads : {256, 512, 1024}) {
        hipMemset(dev, 0, (1 + 256 * 16) * sizeof(unsigned long long));
        hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(threads), 0, 0, dev, 1.0f);
        hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(threads), 0, 0, dev, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(1 + 256 * 16);
        hipMemcpy(h.data(), dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> per;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < threads / 64; ++w) per.push_back((double)h[1 + b * 16 + w] / (64.0 * kIters));
        std::sort(per.begin(), per.end());
        const double med = per[per.size() / 2];
        const int waves_per_simd = threads / 256;
        printf("  %d/SIMD: %6.2f clk/instr/wave = %5.2f clk/instr/SIMD", waves_per_simd, med, med / waves_per_simd);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    unsigned long long* dev = nullptr;
    hipMalloc(&dev, (1 + 256 * 16) * sizeof(unsigned long long));
    printf("gfx950 VALU issue cost, 64-lane waves, %d x 64 instructions per wave (s_memtime clocks)\n", kIters);
    run<ADD>("v_add_f32", dev);
    run<MUL>("v_mul_f32", dev);
    run<FMA>("v_fma_f32", dev);
    run<PK_ADD>("v_pk_add_f32", dev);
    run<PK_MUL>("v_pk_mul_f32", dev);
    run<PK_FMA>("v_pk_fma_f32", dev);
    run<CNDMASK>("v_cndmask vcc", dev);
    run<CNDMASK_SGPR>("v_cndmask sgpr", dev);
    run<PAIR_VCC>("cmp+cnd vcc", dev);
    run<PAIR_SGPR>("cmp+cnd sgpr", dev);
    run<PAIR_SGPR_NOP>("cmp+nop+cnd s", dev);
    run<CMP>("v_cmp_lt_f32", dev);
    run<MAX>("v_max_f32", dev);
    run<AND>("v_and_b32", dev);
    run<ADD_U32>("v_add_u32", dev);
    run<LSHL_ADD>("v_lshl_add_u32", dev);
    run<CVT_I>("v_cvt_i32_f32", dev);
    run<MOV_DPP>("v_mov_dpp", dev);
    run<PK_MOV>("v_pk_mov_b32", dev);
    run<SQRT>("v_sqrt_f32", dev);
    run<RCP>("v_rcp_f32", dev);
    run<ADD_F64>("v_add_f64", dev);
    run<CVT_F64>("v_cvt_f64_f32", dev);
    run<ADD_DEP>("v_add_f32 dep", dev);
    run<PK_ADD_DEP>("v_pk_add dep", dev);
    hipFree(dev);
    return 0;
}
