// issue_microbench.hip -- calibrates the three ceilings the traversal kernel is priced against
// ("replace frac 1.58 with a calibrated ceiling"):
//
//   valu   wave64 vector-ALU instructions per clock per CU at saturation, per opcode class
//          (v_fma_f32, v_pk_fma_f32, v_min3_f32, v_cndmask_b32, v_mov_b32, v_cmp+v_cndmask,
//          v_lshl_add_u64, v_fma_f64) -- "is a wave64 VALU 2 or 4 cycles on gfx950?"
//   salu   scalar-ALU instructions per clock per CU at saturation
//   l1     divergent 16-byte lane loads (global_load_dwordx4, every lane its own 64-byte
//          record) per clock per CU when the table is L1-resident (16 KiB), L2-resident (2 MiB)
//          and Infinity-Cache/HBM-resident (256 MiB); pattern a = 1 load per record,
//          pattern b = the 4 loads of one 64-byte record (k_trace's node fetch),
//          pattern c = quad-cooperative (4 adjacent lanes read the 4 quarters of one record)
//   mix    the k_trace-like mixture: 4 divergent loads + N VALU per iteration, to see whether
//          the two pipes overlap
//
// Every kernel also reports shader-clock cycles (s_memtime) per wave so that rates are per
// CLOCK, independent of the frequency the part actually ran at; wall time (HIP events) gives
// the effective MHz.  Build: hipcc --offload-arch=gfx950 -O3 tools/issue_microbench.hip -o tools/bin/issue_mb
// Under rocprofv3 --pmc it doubles as the calibration of SQ_INSTS_VALU / SQ_INSTS_SALU /
// TCP_TOTAL_CACHE_ACCESSES against KNOWN instruction and access counts (printed per kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// 32 instructions per REP, 8 independent dependency chains (4-deep each) so that one wave alone
// never limits issue; 8 waves per SIMD are resident anyway.
#define REP4(x) x x x x
#define VALU_KERNEL(NAME, BODY)                                                                       \
    __global__ __launch_bounds__(64) void NAME(unsigned iters, float* out, unsigned long long* cyc)   \
    {                                                                                                 \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,      \
              a6 = a0 + 6, a7 = a0 + 7;                                                               \
        float b = 1.0000001f, c = 0.5f;                                                               \
        unsigned long long t0 = __builtin_readcyclecounter();                                        \
        for (unsigned i = 0; i < iters; ++i)                                                          \
        {                                                                                             \
            REP4(BODY)                                                                                \
        }                                                                                             \
        unsigned long long t1 = __builtin_readcyclecounter();                                        \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                   \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                              \
    }

#define ASM8(OP)                                                                                       \
    asm volatile(OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)                              \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                 : "v"(b), "v"(c));

#define OP_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define OP_MUL(r) "v_mul_f32 " #r ", " #r ", %8\n"
#define OP_ADD(r) "v_add_f32 " #r ", " #r ", %9\n"
#define OP_MIN3(r) "v_min3_f32 " #r ", " #r ", %8, %9\n"
#define OP_MAX(r) "v_max_f32 " #r ", " #r ", %8\n"
#define OP_MOV(r) "v_mov_b32 " #r ", %8\n"
#define OP_CND(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc\n"
#define OP_CMPCND(r) "v_cmp_lt_f32 vcc, " #r ", %8\n v_cndmask_b32 " #r ", " #r ", %9, vcc\n"
#define OP_AND(r) "v_and_b32 " #r ", " #r ", %8\n"

VALU_KERNEL(k_fma, ASM8(OP_FMA))
VALU_KERNEL(k_mul, ASM8(OP_MUL))
VALU_KERNEL(k_add, ASM8(OP_ADD))
VALU_KERNEL(k_min3, ASM8(OP_MIN3))
VALU_KERNEL(k_max, ASM8(OP_MAX))
VALU_KERNEL(k_mov, ASM8(OP_MOV))
VALU_KERNEL(k_cnd, ASM8(OP_CND))
VALU_KERNEL(k_cmpcnd, ASM8(OP_CMPCND))     // 64 instructions per REP4
VALU_KERNEL(k_and, ASM8(OP_AND))
// round 3: the opcodes loop C of k_trace_w4 is made of, to see which of its selects / conversions / tests have a cheaper form
#define OP_BFI(r) "v_bfi_b32 " #r ", %8, " #r ", %9\n"
#define OP_XOR(r) "v_xor_b32 " #r ", " #r ", %8\n"
#define OP_BFEI(r) "v_bfe_i32 " #r ", " #r ", 2, 5\n"
#define OP_BFEU(r) "v_bfe_u32 " #r ", " #r ", 2, 5\n"
#define OP_CVTUB(r) "v_cvt_f32_ubyte1 " #r ", " #r "\n"
#define OP_CVTU(r) "v_cvt_f32_u32 " #r ", " #r "\n"
#define OP_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9\n"
#define OP_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9\n"
#define OP_MED3(r) "v_med3_f32 " #r ", " #r ", %8, %9\n"
#define OP_CMPS(r) "v_cmp_lt_f32 s[20:21], " #r ", %8\n"
#define OP_CNDS(r) "v_cndmask_b32 " #r ", " #r ", %8, s[22:23]\n"
#define OP_LSHR(r) "v_lshrrev_b32 " #r ", 3, " #r "\n"
#define OP_ANDOR(r) "v_and_or_b32 " #r ", " #r ", %8, %9\n"
#define OP_FMAC(r) "v_fmac_f32 " #r ", %8, %9\n"
#define ASM8C(OP)                                                                                      \
    asm volatile(OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)                              \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                 : "v"(b), "v"(c) : "s20", "s21");
#define ASM8M(OP)                                                                                      \
    asm volatile("s_mov_b32 s22, 0x55555555\n s_mov_b32 s23, 0x55555555\n" OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                 : "v"(b), "v"(c) : "s22", "s23");
VALU_KERNEL(k_bfi, ASM8(OP_BFI))
VALU_KERNEL(k_xor, ASM8(OP_XOR))
VALU_KERNEL(k_bfei, ASM8(OP_BFEI))
VALU_KERNEL(k_bfeu, ASM8(OP_BFEU))
VALU_KERNEL(k_cvtub, ASM8(OP_CVTUB))
VALU_KERNEL(k_cvtu, ASM8(OP_CVTU))
VALU_KERNEL(k_perm, ASM8(OP_PERM))
VALU_KERNEL(k_max3, ASM8(OP_MAX3))
VALU_KERNEL(k_med3, ASM8(OP_MED3))
VALU_KERNEL(k_cmps, ASM8C(OP_CMPS))
VALU_KERNEL(k_cnds, ASM8M(OP_CNDS))        // (+ one s_mov per 8)
VALU_KERNEL(k_lshr, ASM8(OP_LSHR))
VALU_KERNEL(k_andor, ASM8(OP_ANDOR))
VALU_KERNEL(k_fmac, ASM8(OP_FMAC))

// packed fp32: two floats per lane per instruction
__global__ __launch_bounds__(64) void k_pkfma(unsigned iters, float* out, unsigned long long* cyc)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
       a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {1.0000001f, 1.0000002f}, c = {0.5f, 0.25f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < iters; ++i)
    {
#define OP_PK(r) "v_pk_fma_f32 " #r ", " #r ", %8, %9\n"
        REP4(ASM8(OP_PK))
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(64) void k_fma64(unsigned iters, float* out, unsigned long long* cyc)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001, c = 0.5;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < iters; ++i)
    {
#define OP_F64(r) "v_fma_f64 " #r ", " #r ", %8, %9\n"
        REP4(ASM8(OP_F64))
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(64) void k_lshladd64(unsigned iters, float* out, unsigned long long* cyc)
{
    unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long b = 12345, c = 3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < iters; ++i)
    {
#define OP_LA(r) "v_lshl_add_u64 " #r ", " #r ", 1, %8\n"
        REP4(ASM8(OP_LA))
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// scalar ALU: 32 s_add/s_xor per REP4 on 8 independent SGPR chains
__global__ __launch_bounds__(64) void k_salu(unsigned iters, float* out, unsigned long long* cyc)
{
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < iters; ++i)
    {
#define OP_S(r) "s_add_u32 " #r ", " #r ", 3\n"
#define SASM8 asm volatile(OP_S(%0) OP_S(%1) OP_S(%2) OP_S(%3) OP_S(%4) OP_S(%5) OP_S(%6) OP_S(%7) \
                           : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");
        REP4(SASM8)
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// divergent record loads.  The index stream is an LCG per lane (not data dependent): this is a
// throughput test, 8 waves per SIMD keep the pipe full.
//   PATTERN 0: one dwordx4 per lane per step, every lane a different 64-byte record
//   PATTERN 1: the four dwordx4 of one record per lane per step (k_trace's node fetch)
//   PATTERN 2: quad-cooperative: lanes 4q..4q+3 read the four quarters of ONE record per load,
//              four loads per step cover four records per quad (same bytes per lane as pattern 1)
//   VALU_PER_STEP: extra dependent-free v_fma_f32 per step (the "mix" runs)
#define USE4(q) acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
template <int PATTERN, int VALU_PER_STEP>
__global__ __launch_bounds__(64) void k_loads(const float4* __restrict__ recs, unsigned mask, unsigned iters, float* out,
    unsigned long long* cyc)
{
    const unsigned lane = threadIdx.x;
    unsigned s = (blockIdx.x * 64u + lane) * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    float f0 = lane, f1 = lane + 1.f, f2 = lane + 2.f, f3 = lane + 3.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned it = 0; it < iters; ++it)
    {
        s = s * 1664525u + 1013904223u;
        if (PATTERN == 0)
        {
            float4 q = recs[(size_t)((s >> 8) & mask) * 4 + (s & 3u)];
            acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
        }
        else if (PATTERN == 1)
        {
            const float4* p = recs + (size_t)((s >> 8) & mask) * 4;
            float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            USE4(q0) USE4(q1) USE4(q2) USE4(q3)
        }
        else
        {
            // the quad leader's four record indices, broadcast inside the quad (DPP quad_perm)
            unsigned s_q = __builtin_amdgcn_mov_dpp(s, 0x00, 0xF, 0xF, true);     // quad_perm [0,0,0,0]
            unsigned i0 = (s_q >> 8) & mask, i1 = (s_q >> 9) & mask, i2 = (s_q >> 10) & mask, i3 = (s_q >> 11) & mask;
            float4 q0 = recs[(size_t)i0 * 4 + (lane & 3u)], q1 = recs[(size_t)i1 * 4 + (lane & 3u)],
                   q2 = recs[(size_t)i2 * 4 + (lane & 3u)], q3 = recs[(size_t)i3 * 4 + (lane & 3u)];
            USE4(q0) USE4(q1) USE4(q2) USE4(q3)
        }
#pragma unroll
        for (int k = 0; k < VALU_PER_STEP / 4; ++k)
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0000001f), "v"(0.5f));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w + f0 + f1 + f2 + f3;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Result { double ms, cycles; };

template <class F>
static Result timed(F launch, unsigned blocks, unsigned long long* d_cyc)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();                                   // warm-up
    (void)hipEventRecord(a);
    launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> c(blocks);
    (void)hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return {ms, mean / blocks};
}

int main(int argc, char** argv)
{
    const char* only = argc > 1 ? argv[1] : "all";
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const unsigned cus = prop.multiProcessorCount;
    const unsigned waves_per_cu = 32;                       // 8 per SIMD: full residency
    const unsigned blocks = cus * waves_per_cu;
    printf("device %s, %u CUs, clock %d kHz, %u one-wave blocks (%u per CU)\n", prop.gcnArchName, cus, prop.clockRate, blocks,
        waves_per_cu);
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * 4));
    CHECK(hipMalloc(&cyc, (size_t)blocks * 8));
    auto want = [&](const char* n) { return !strcmp(only, "all") || !strcmp(only, n); };

    // ---- vector / scalar ALU issue ------------------------------------------------------
    if (want("valu") || want("salu"))
    {
        const unsigned iters = 20000;
        struct K { const char* name; void (*fn)(unsigned, float*, unsigned long long*); double per_iter; bool scalar; };
        K ks[] = {
            {"v_fma_f32", k_fma, 32, false}, {"v_mul_f32", k_mul, 32, false}, {"v_add_f32", k_add, 32, false},
            {"v_min3_f32", k_min3, 32, false}, {"v_max_f32", k_max, 32, false}, {"v_mov_b32", k_mov, 32, false},
            {"v_cndmask_b32", k_cnd, 32, false}, {"v_cmp_lt_f32+v_cndmask_b32", k_cmpcnd, 64, false},
            {"v_and_b32", k_and, 32, false}, {"v_pk_fma_f32", k_pkfma, 32, false}, {"v_fma_f64", k_fma64, 32, false},
            {"v_lshl_add_u64", k_lshladd64, 32, false}, {"s_add_u32", k_salu, 32, true},
            {"v_bfi_b32", k_bfi, 32, false}, {"v_xor_b32", k_xor, 32, false}, {"v_bfe_i32", k_bfei, 32, false}, {"v_bfe_u32", k_bfeu, 32, false},
            {"v_cvt_f32_ubyte1", k_cvtub, 32, false}, {"v_cvt_f32_u32", k_cvtu, 32, false}, {"v_perm_b32", k_perm, 32, false},
            {"v_max3_f32", k_max3, 32, false}, {"v_med3_f32", k_med3, 32, false}, {"v_cmp_lt_f32 -> sgpr pair", k_cmps, 32, false},
            {"v_cndmask_b32 (sgpr mask)", k_cnds, 32, false}, {"v_lshrrev_b32", k_lshr, 32, false}, {"v_and_or_b32", k_andor, 32, false},
            {"v_fmac_f32", k_fmac, 32, false},
        };
        for (auto& k : ks)
        {
            if (k.scalar ? !want("salu") : !want("valu")) continue;
            Result r = timed([&]() { hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, iters, out, cyc); }, blocks, cyc);
            double insts = (double)blocks * iters * k.per_iter;              // wave-level instructions
            double per_clk_cu = insts / cus / r.cycles;                      // all waves of a CU run concurrently for r.cycles
            printf("%-28s %8.3f ms  %.0f cycles/wave  eff. clock %.0f MHz | %.3f wave-instr/clk/CU = %.3f /clk/SIMD -> %.2f cycles per wave64 instruction | known wave-instructions %.4g\n",
                k.name, r.ms, r.cycles, r.cycles / r.ms / 1e3, per_clk_cu, per_clk_cu / 4, 4.0 / per_clk_cu, insts);
        }
    }

    // ---- divergent loads ------------------------------------------------------------------
    if (want("l1") || want("mix"))
    {
        const size_t max_recs = (size_t)1 << 22;                               // 256 MiB of 64-byte records
        float4* recs;
        CHECK(hipMalloc(&recs, max_recs * 64));
        CHECK(hipMemset(recs, 0, max_recs * 64));
        struct T { const char* name; unsigned mask; };
        T tables[] = {{"16 KiB (L1)", (1u << 8) - 1}, {"2 MiB (L2)", (1u << 15) - 1}, {"256 MiB (MALL/HBM)", (1u << 22) - 1}};
        const unsigned iters = 4000;
        for (auto& t : tables)
        {
            auto report = [&](const char* pat, Result r, double lane_loads, double accesses, double valu)
            {
                printf("%-20s %-34s %8.3f ms  %.0f cycles/wave | %.3f lane-loads(16 B)/clk/CU = %.1f B/clk/CU | expected L1 accesses %.4g (%.3f /clk/CU)",
                    t.name, pat, r.ms, r.cycles, lane_loads / cus / r.cycles, lane_loads * 16 / cus / r.cycles, accesses,
                    accesses / cus / r.cycles);
                if (valu > 0) printf(" | + %.3f VALU/clk/CU", valu / cus / r.cycles);
                printf("\n");
            };
            double n = (double)blocks * 64 * iters;
            if (want("l1"))
            {
                Result r = timed([&]() { hipLaunchKernelGGL((k_loads<0, 0>), dim3(blocks), dim3(64), 0, 0, recs, t.mask, iters, out, cyc); }, blocks, cyc);
                report("a: 1 x dwordx4, lanes distinct", r, n, n, 0);
                r = timed([&]() { hipLaunchKernelGGL((k_loads<1, 0>), dim3(blocks), dim3(64), 0, 0, recs, t.mask, iters, out, cyc); }, blocks, cyc);
                report("b: 4 x dwordx4 of one record", r, 4 * n, 4 * n, 0);
                r = timed([&]() { hipLaunchKernelGGL((k_loads<2, 0>), dim3(blocks), dim3(64), 0, 0, recs, t.mask, iters, out, cyc); }, blocks, cyc);
                report("c: quad-cooperative 4 x dwordx4", r, 4 * n, n, 0);
            }
            if (want("mix"))
            {
                Result r = timed([&]() { hipLaunchKernelGGL((k_loads<1, 32>), dim3(blocks), dim3(64), 0, 0, recs, t.mask, iters, out, cyc); }, blocks, cyc);
                report("b + 32 v_fma per step", r, 4 * n, 4 * n, (double)blocks * iters * 32);
                r = timed([&]() { hipLaunchKernelGGL((k_loads<1, 64>), dim3(blocks), dim3(64), 0, 0, recs, t.mask, iters, out, cyc); }, blocks, cyc);
                report("b + 64 v_fma per step", r, 4 * n, 4 * n, (double)blocks * iters * 64);
                r = timed([&]() { hipLaunchKernelGGL((k_loads<1, 128>), dim3(blocks), dim3(64), 0, 0, recs, t.mask, iters, out, cyc); }, blocks, cyc);
                report("b + 128 v_fma per step", r, 4 * n, 4 * n, (double)blocks * iters * 128);
                r = timed([&]() { hipLaunchKernelGGL((k_loads<2, 64>), dim3(blocks), dim3(64), 0, 0, recs, t.mask, iters, out, cyc); }, blocks, cyc);
                report("c + 64 v_fma per step", r, 4 * n, n, (double)blocks * iters * 64);
            }
        }
    }
    return 0;
}
