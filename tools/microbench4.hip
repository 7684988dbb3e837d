// Issue cost of v_pk_fma_f32 on gfx950 by operand pattern (cycles per wave-instruction per
// SIMD at 4 waves/SIMD; nominal clock from argv[1]):
//   0  all three sources VGPR pairs, three different ones
//   1  one source an SGPR pair
//   2  one source an SGPR pair with op_sel_hi:[0,1,1] (low half broadcast to both lanes)
//   3  v_pk_mul_f32 with an SGPR pair
//   4  plain v_fma_f32 with an SGPR source (reference)
//   5  sources VGPR pairs, accumulator also a multiplicand (two different register pairs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int V>
__global__ void __launch_bounds__(256) k(float* out, const float* coef, int iters) {
    const float c0 = coef[0], c1 = coef[1];
    v2f a[8], b[8];
    float f[8];
    for (int i = 0; i < 8; i++) { a[i].x = threadIdx.x * 0.001f + i; a[i].y = a[i].x + 0.5f; b[i] = a[i] * 0.5f; f[i] = a[i].x; }
    const v2f k1 = {c1, c1};
    unsigned long long sp;
    {
        unsigned int lo = __builtin_amdgcn_readfirstlane(__float_as_int(c0));
        sp = ((unsigned long long)lo << 32) | lo;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (V == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(k1));
                if (V == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sp), "v"(k1));
                if (V == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(sp), "v"(k1));
                if (V == 3) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sp));
                if (V == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "s"(c0), "v"(c1));
                if (V == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(k1));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y + f[i] + b[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V> void run(int cus, float* out, float* coef, double ghz, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, wps = 4;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(cus * wps), dim3(256), 0, 0, out, coef, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr = (double)iters * 64 * wps, cycles = ms * 1e-3 * ghz * 1e9;
    printf("%d %-60s %.3f ms  %.2f cycles per instruction per SIMD\n", V, name, ms, cycles / instr);
}

int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out, *coef; hipMalloc(&out, (size_t)256 * cus * 8 * 4); hipMalloc(&coef, 8);
    float hc[2] = {0.999f, 0.001f}; hipMemcpy(coef, hc, 8, hipMemcpyHostToDevice);
    run<4>(cus, out, coef, ghz, "v_fma_f32 v, s, v, v (warm-up)");
    run<0>(cus, out, coef, ghz, "v_pk_fma_f32 v, v, v, v (three register pairs)");
    run<1>(cus, out, coef, ghz, "v_pk_fma_f32 v, s, v, v");
    run<2>(cus, out, coef, ghz, "v_pk_fma_f32 v, s, v, v op_sel_hi:[0,1,1]");
    run<3>(cus, out, coef, ghz, "v_pk_mul_f32 v, s, v");
    run<4>(cus, out, coef, ghz, "v_fma_f32 v, s, v, v");
    run<5>(cus, out, coef, ghz, "v_pk_fma_f32 v, v, k, v (two register pairs)");
    return 0;
}
