// pk_fma_coissue_probe.hip -- standalone check of what tools/hazard_probe.py found inside the engine:
// v_pk_fma_f32 with cross-half operand selection (op_sel / op_sel_hi) losing ONE product of ONE half in a 16-lane group while waves of
// another kernel issue bf16 MFMAs on the same SIMDs; v_fmac_f32 on the same registers at the same time is right.
//
// Victim: every thread holds five tap pairs T_n = (cr, ci) and five history pairs L_n = (sx, sy) in registers and forms
//   A = (sum sx ci, sum sy ci), B = (sum sx cr, sum sy cr)        (deep filter: re = B.lo - A.hi, im = B.hi + A.lo)
// twice per iteration: with the ten packed instructions of the failing df_apply build (same operand selections, in-place accumulation,
// the no-op s_waitcnt between the pairs) and with twenty scalar v_mul / v_fmac.  Both are fused multiply-adds in the same order: bit-equal.
// A mismatch is counted and the first few are recorded (which half, which lanes).
// Aggressor (second stream, one 256-thread workgroup per CU, runs until told to stop): 0 none | 1 v_mfma_f32_16x16x32_bf16 chains |
// 2 v_mfma_f32_16x16x4_f32 chains | 3 bf16 MFMA chains fed by ds_read_b128 with v_cvt_pk_bf16_f32 + exp2 / rcp gate math between (limb-like).
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/pk_fma_coissue_probe tools/pk_fma_coissue_probe.hip
// run:   tools/pk_fma_coissue_probe [seconds per aggressor kind = 5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void aggressor_kernel(int kind, volatile int* stop, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[3][16][72];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * 16 * 72; i += 256) ((unsigned short*)lds)[i] = (unsigned short)(0x3c00 + (i & 63));
    __syncthreads();
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f64x4 dacc[4] = {{0., 0., 0., 0.}, {0., 0., 0., 0.}, {0., 0., 0., 0.}, {0., 0., 0., 0.}};
    f32x16 wacc[2] = {};
    uint4 a = {0x3f803f80u + lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}, b = {0x3f803f80u, 0x3f003f00u + lane, 0x3e803e80u, 0x3f003f00u};
    float fa = 1.0f + lane * 1e-3f, fb = 0.5f;
    long it = 0;
    while (!*stop && it < (1L << 22)) {        // (capped: a host that died must not leave the GPU spinning)
        for (int r = 0; r < 64; ++r) {
            if (kind == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[k], 0, 0, 0);
            } else if (kind == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[k], 0, 0, 0);
            } else if (kind == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) dacc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa, (double)fb, dacc[k], 0, 0, 0);
            } else if (kind == 5) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, acc[k], 0, 0, 0);
            } else if (kind == 6) {
#pragma unroll
                for (int k = 0; k < 2; ++k) wacc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), wacc[k], 0, 0, 0);
            } else if (kind == 7) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc[k], 0, 0, 0);
            } else if (kind == 8) {       // no matrix instructions at all: packed bf16 conversions + transcendentals (the limb kernels' VALU side)
                typedef float f32x2_t __attribute__((ext_vector_type(2))); typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const f32x2_t v = {fa, fb};
                const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                fa = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fb * 1e-3f)) + (float)(p & 1);
                fb = fb * 0.999f + 1e-3f;
            } else if (kind == 3) {
                uint4 hb[3];
#pragma unroll
                for (int l = 0; l < 3; ++l) hb[l] = *(const uint4*)&lds[l][lane & 15][8 * (lane >> 4)];
#pragma unroll
                for (int l = 0; l < 3; ++l)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, hb[l]), acc[k], 0, 0, 0);
                float g = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0][0] * 1e-6f));
                typedef float f32x2_t __attribute__((ext_vector_type(2))); typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const f32x2_t v = {g, acc[1][1]};
                const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                *(unsigned*)&lds[r % 3][lane & 15][2 * (lane >> 4)] = p | 0x3c003c00u;
                acc[0][0] = g;
            }
        }
        ++it;
    }
    if (sink) sink[blockIdx.x * 256 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)it + (float)dacc[0][0] + (float)dacc[3][1] + wacc[0][3] + wacc[1][7] + fa;
}

// data: [thread][20] floats = T0..T4 (cr, ci), L0..L4 (sx, sy).  detail: up to 64 records of {global thread, iteration, lane, packed A.lo, A.hi, B.lo, B.hi, twin ...}
__global__ __launch_bounds__(256) void victim_kernel(const float* data, int iters, unsigned* nbad, unsigned* bad_half, float* detail) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float2* p = (const float2*)(data + gid * 20);
    float2 T0 = p[0], T1 = p[1], T2 = p[2], T3 = p[3], T4 = p[4], L0 = p[5], L1 = p[6], L2 = p[7], L3 = p[8], L4 = p[9];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(T0), "+v"(T1), "+v"(T2), "+v"(T3), "+v"(T4), "+v"(L0), "+v"(L1), "+v"(L2), "+v"(L3), "+v"(L4));
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        float2 A, B, A2;
        float s32, s33, s34, s35;
        asm volatile(
            "v_pk_fma_f32 %[A], %[T0], %[L0], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 %[B], %[T0], %[L0], 0 op_sel_hi:[0,1,0]\n\t"
            "s_waitcnt vmcnt(3)\n\t"
            "v_pk_fma_f32 %[A], %[L1], %[T1], %[A] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[B], %[L1], %[T1], %[B] op_sel_hi:[1,0,1]\n\t"
            "s_waitcnt vmcnt(2)\n\t"
            "v_pk_fma_f32 %[A], %[L2], %[T2], %[A] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[B], %[L2], %[T2], %[B] op_sel_hi:[1,0,1]\n\t"
            "s_waitcnt vmcnt(1)\n\t"
            "v_pk_fma_f32 %[A], %[L3], %[T3], %[A] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[B], %[L3], %[T3], %[B] op_sel_hi:[1,0,1]\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            "v_pk_fma_f32 %[A2], %[L4], %[T4], %[A] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[B], %[L4], %[T4], %[B] op_sel_hi:[1,0,1]\n\t"
            "s_nop 1"
            : [A] "=&v"(A), [B] "=&v"(B), [A2] "=&v"(A2)
            : [T0] "v"(T0), [T1] "v"(T1), [T2] "v"(T2), [T3] "v"(T3), [T4] "v"(T4), [L0] "v"(L0), [L1] "v"(L1), [L2] "v"(L2), [L3] "v"(L3), [L4] "v"(L4));
        // scalar twins (asm: the compiler would pack them): s32 = sum sx ci (A.lo), s33 = sum sy ci (A.hi), s34 = sum sx cr (B.lo), s35 = sum sy cr (B.hi)
        float t0x = T0.x, t0y = T0.y, t1x = T1.x, t1y = T1.y, t2x = T2.x, t2y = T2.y, t3x = T3.x, t3y = T3.y, t4x = T4.x, t4y = T4.y;
        float l0x = L0.x, l0y = L0.y, l1x = L1.x, l1y = L1.y, l2x = L2.x, l2y = L2.y, l3x = L3.x, l3y = L3.y, l4x = L4.x, l4y = L4.y;
        asm volatile(
            "v_mul_f32 %0, %5, %14\n\tv_mul_f32 %1, %5, %15\n\tv_mul_f32 %2, %4, %14\n\tv_mul_f32 %3, %4, %15\n\t"
            "v_fmac_f32 %0, %16, %7\n\tv_fmac_f32 %1, %17, %7\n\tv_fmac_f32 %2, %16, %6\n\tv_fmac_f32 %3, %17, %6\n\t"
            "v_fmac_f32 %0, %18, %9\n\tv_fmac_f32 %1, %19, %9\n\tv_fmac_f32 %2, %18, %8\n\tv_fmac_f32 %3, %19, %8\n\t"
            "v_fmac_f32 %0, %20, %11\n\tv_fmac_f32 %1, %21, %11\n\tv_fmac_f32 %2, %20, %10\n\tv_fmac_f32 %3, %21, %10\n\t"
            "v_fmac_f32 %0, %22, %13\n\tv_fmac_f32 %1, %23, %13\n\tv_fmac_f32 %2, %22, %12\n\tv_fmac_f32 %3, %23, %12"
            : "=&v"(s32), "=&v"(s33), "=&v"(s34), "=&v"(s35)
            : "v"(t0x), "v"(t0y), "v"(t1x), "v"(t1y), "v"(t2x), "v"(t2y), "v"(t3x), "v"(t3y), "v"(t4x), "v"(t4y),
              "v"(l0x), "v"(l0y), "v"(l1x), "v"(l1y), "v"(l2x), "v"(l2y), "v"(l3x), "v"(l3y), "v"(l4x), "v"(l4y));
        const unsigned m = (__float_as_uint(A2.x) != __float_as_uint(s32) ? 1u : 0u) | (__float_as_uint(A2.y) != __float_as_uint(s33) ? 2u : 0u) |
                           (__float_as_uint(B.x) != __float_as_uint(s34) ? 4u : 0u) | (__float_as_uint(B.y) != __float_as_uint(s35) ? 8u : 0u);
        if (m) {
            ++bad;
            const unsigned k = atomicAdd(nbad, 1u);
            for (int h = 0; h < 4; ++h) if (m & (1u << h)) atomicAdd(bad_half + h, 1u);
            if (k < 64) {
                float* d = detail + k * 12;
                d[0] = (float)gid; d[1] = (float)it; d[2] = (float)(threadIdx.x & 63); d[3] = (float)m;
                d[4] = A2.x; d[5] = A2.y; d[6] = B.x; d[7] = B.y; d[8] = s32; d[9] = s33; d[10] = s34; d[11] = s35;
            }
        }
        // keep the operands opaque so that the loop is not hoisted
        asm volatile("" : "+v"(T0), "+v"(T1), "+v"(T2), "+v"(T3), "+v"(T4), "+v"(L0), "+v"(L1), "+v"(L2), "+v"(L3), "+v"(L4));
    }
    if (bad == 0xffffffffu) detail[0] = 1.f;
}

// One packed instruction form at a time: D = packed op of a = (a0, a1), b = (b0, b1), c = (c0, c1) against the two scalar instructions it stands for.
//   form 0 fma default            lo = a0 b0 + c0, hi = a1 b1 + c1          form 1 fma op_sel:[0,1,0]     lo = a0 b1 + c0
//   form 2 fma op_sel:[1,0,0]     lo = a1 b0 + c0                           form 3 fma op_sel_hi:[1,0,1]  hi = a1 b0 + c1
//   form 4 fma op_sel_hi:[0,1,1]  hi = a0 b1 + c1                           form 5 mul op_sel:[0,1]       lo = a0 b1
//   form 6 add op_sel:[0,1]       lo = a0 + b1                              form 7 fma op_sel:[0,1,0], accumulating IN PLACE (D = c)
//   form 8 fma op_sel:[0,0,1]     lo = a0 b0 + c1                           form 9 pk_mov op_sel:[1,0]    (lo, hi) = (a1, b0)
template <int FORM>
__global__ __launch_bounds__(256) void form_kernel(const float* data, int iters, unsigned* nbad_lo, unsigned* nbad_hi) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float2* p = (const float2*)(data + gid * 20);
    float2 a = p[0], b = p[1], c = p[2];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
    for (int it = 0; it < iters; ++it) {
        float2 d; float lo, hi;
        float a0 = a.x, a1 = a.y, b0 = b.x, b1 = b.y, c0 = c.x, c1 = c.y;
        if (FORM == 0) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
                         asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)); }
        if (FORM == 1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
                         asm volatile("v_fma_f32 %0, %2, %5, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)); }
        if (FORM == 2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
                         asm volatile("v_fma_f32 %0, %3, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)); }
        if (FORM == 3) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
                         asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %4, %7" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)); }
        if (FORM == 4) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
                         asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %2, %5, %7" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)); }
        if (FORM == 5) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b));
                         asm volatile("v_mul_f32 %0, %2, %5\n\tv_mul_f32 %1, %3, %5" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1)); }
        if (FORM == 6) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b));
                         asm volatile("v_add_f32 %0, %2, %5\n\tv_add_f32 %1, %3, %5" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1)); }
        if (FORM == 7) { d = c; asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]\n\ts_nop 1" : "+v"(d) : "v"(a), "v"(b));
                         asm volatile("v_fma_f32 %0, %2, %5, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)); }
        if (FORM == 8) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
                         asm volatile("v_fma_f32 %0, %2, %4, %7\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)); }
        if (FORM == 9) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b)); lo = a1; hi = b1; }
        if (__float_as_uint(d.x) != __float_as_uint(lo)) atomicAdd(nbad_lo, 1u);
        if (__float_as_uint(d.y) != __float_as_uint(hi)) atomicAdd(nbad_hi, 1u);
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    }
}
template <int FORM>
static void launch_form(hipStream_t sv, int wgs, const float* data, int iters, unsigned* lo, unsigned* hi) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(form_kernel<FORM>), dim3(wgs), dim3(256), 0, sv, data, iters, lo, hi);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 5.0;
    const int victim_wgs = 8192, iters = 64;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *data, *sink, *detail; unsigned *nbad, *bad_half; int* stop;
    const size_t nthr = (size_t)victim_wgs * 256;
    CK(hipMalloc(&data, nthr * 20 * 4)); CK(hipMalloc(&sink, (size_t)cus * 4 * 256 * 4)); CK(hipMalloc(&detail, 64 * 12 * 4));
    CK(hipMalloc(&nbad, 4)); CK(hipMalloc(&bad_half, 16));
    CK(hipHostMalloc(&stop, 4, hipHostMallocDefault));
    std::vector<float> h(nthr * 20);
    unsigned long long sd = 88172645463325252ull;
    for (auto& v : h) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v = ((float)((sd >> 11) & 0xfffff) / 1048576.f * 2.f - 1.f) * 0.05f; }
    CK(hipMemcpy(data, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sv; CK(hipStreamCreate(&sa)); CK(hipStreamCreateWithPriority(&sv, hipStreamNonBlocking, -1));
    const char* names[] = {"no aggressor", "bf16 MFMA 16x16x32 chains", "fp32 MFMA 16x16x4 chains", "bf16 MFMA + ds_read_b128 + cvt_pk_bf16 + exp2/rcp (limb-like)",
                           "f64 MFMA 16x16x4 chains", "fp32 MFMA 4x4x1 chains", "bf16 MFMA 32x32x16 chains", "f16 MFMA 16x16x32 chains",
                           "no MFMA: cvt_pk_bf16 + exp2 + rcp"};
    const int order[] = {0, 1, 2, 3, 4, 6, 7, 8};       // (5, the 4x4x1 shape, is left out: the probe did not come back from it once)
    for (int occ = 1; occ <= 2; ++occ)
    for (int oi = 0; oi < 8; ++oi) {
        const int kind = order[oi];
        if (occ == 2 && (kind == 0 || kind > 3)) continue;
        CK(hipMemset(nbad, 0, 4)); CK(hipMemset(bad_half, 0, 16)); *stop = 0;
        if (kind) hipLaunchKernelGGL(aggressor_kernel, dim3(cus * occ), dim3(256), 0, sa, kind, stop, sink);
        const auto t0 = std::chrono::steady_clock::now();
        long launches = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
            for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(victim_kernel, dim3(victim_wgs), dim3(256), 0, sv, data, iters, nbad, bad_half, detail);
            CK(hipStreamSynchronize(sv)); launches += 8;
        }
        *stop = 1;
        CK(hipDeviceSynchronize());
        unsigned nb = 0, bh[4]; CK(hipMemcpy(&nb, nbad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bh, bad_half, 16, hipMemcpyDeviceToHost));
        const double evals = (double)launches * nthr * iters;
        printf("%-66s x %d WG/CU: %ld victim launches, %.3g packed sums checked: %u mismatches (A.lo %u, A.hi %u, B.lo %u, B.hi %u)\n", names[kind], occ, launches, evals, nb, bh[0], bh[1], bh[2], bh[3]);
        if (nb) {
            std::vector<float> d(64 * 12); CK(hipMemcpy(d.data(), detail, d.size() * 4, hipMemcpyDeviceToHost));
            for (unsigned k = 0; k < (nb < 8 ? nb : 8); ++k)
                printf("    thread %.0f iteration %.0f lane %.0f halves %x: packed (%.9g %.9g | %.9g %.9g) scalar (%.9g %.9g | %.9g %.9g)\n", d[k * 12], d[k * 12 + 1], d[k * 12 + 2],
                       (unsigned)d[k * 12 + 3], d[k * 12 + 4], d[k * 12 + 5], d[k * 12 + 6], d[k * 12 + 7], d[k * 12 + 8], d[k * 12 + 9], d[k * 12 + 10], d[k * 12 + 11]);
        }
        fflush(stdout);
    }
    if (argc < 3) return 0;
    // ---- (second argument given) single packed instructions of several forms: NONE of them fails on its own -- the chain above does
    const char* knames[] = {"none", "bf16 16x16x32", "fp32 16x16x4", "limb-like", "f64 16x16x4", "fp32 4x4x1", "bf16 32x32x16", "f16 16x16x32", "no MFMA: cvt_pk_bf16 + exp2 + rcp"};
    const char* fnames[] = {"fma default", "fma op_sel:[0,1,0]", "fma op_sel:[1,0,0]", "fma op_sel_hi:[1,0,1]", "fma op_sel_hi:[0,1,1]", "mul op_sel:[0,1]", "add op_sel:[0,1]",
                            "fma op_sel:[0,1,0] in place", "fma op_sel:[0,0,1]", "pk_mov op_sel:[1,0]"};
    const double fsecs = secs / 4 > 0.5 ? secs / 4 : 0.5;
    printf("\nform scan: mismatching (lo | hi) results per 1e9 packed instructions\n%-36s", "aggressor \\ packed form");
    for (int f = 0; f < 10; ++f) printf(" | f%d", f);
    printf("\n");
    for (int kind = 0; kind < 5; ++kind) {
        *stop = 0;
        if (kind) hipLaunchKernelGGL(aggressor_kernel, dim3(cus), dim3(256), 0, sa, kind, stop, sink);
        printf("%-36s", knames[kind]);
        for (int f = 0; f < 10; ++f) {
            CK(hipMemset(nbad, 0, 4)); CK(hipMemset(bad_half, 0, 16));
            const auto t0 = std::chrono::steady_clock::now();
            long launches = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < fsecs) {
                for (int k = 0; k < 4; ++k) {
                    switch (f) {
                    case 0: launch_form<0>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break; case 1: launch_form<1>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break;
                    case 2: launch_form<2>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break; case 3: launch_form<3>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break;
                    case 4: launch_form<4>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break; case 5: launch_form<5>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break;
                    case 6: launch_form<6>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break; case 7: launch_form<7>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break;
                    case 8: launch_form<8>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break; default: launch_form<9>(sv, victim_wgs, data, iters, bad_half, bad_half + 1); break;
                    }
                }
                CK(hipStreamSynchronize(sv)); launches += 4;
            }
            unsigned bh[2]; CK(hipMemcpy(bh, bad_half, 8, hipMemcpyDeviceToHost));
            const double per = 1e9 / ((double)launches * nthr * iters);
            printf(" | %.3g %.3g", bh[0] * per, bh[1] * per);
            fflush(stdout);
        }
        printf("\n");
        *stop = 1;
        CK(hipDeviceSynchronize());
    }
    for (int f = 0; f < 10; ++f) printf("  f%d = %s\n", f, fnames[f]);
    return 0;
}
