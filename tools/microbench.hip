// VALU issue-rate microbenchmark for gfx950: how many wave64 FP32 instructions per cycle
// does one SIMD retire, as a function of waves per SIMD, for v_fma_f32 with a scalar
// operand, v_pk_fma_f32, and v_rcp_f32?  Prints instructions per cycle per SIMD assuming
// the clock given on the command line (default 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, const float* coef, int iters) {
    float c0 = coef[0], c1 = coef[1];     // uniform -> SGPR
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    if (KIND == 0) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = __builtin_fmaf(a[i], c0, c1);
        }
    } else if (KIND == 1) {
        v2f p[4];
        for (int i = 0; i < 4; i++) { p[i].x = a[2 * i]; p[i].y = a[2 * i + 1]; }
        v2f m = {c0, c0}, ad = {c1, c1};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++) p[i] = __builtin_elementwise_fma(p[i], m, ad);
        }
        for (int i = 0; i < 4; i++) { a[2 * i] = p[i].x; a[2 * i + 1] = p[i].y; }
    } else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = __builtin_amdgcn_rcpf(a[i]);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float *out, *coef; hipMalloc(&out, 256 * cus * 8 * 256 * 4); hipMalloc(&coef, 8);
    float hc[2] = {0.999f, 0.001f}; hipMemcpy(coef, hc, 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[3] = {"v_fma_f32 (sgpr operand)", "v_pk_fma_f32", "v_rcp_f32"};
    for (int kind = 0; kind < 3; kind++)
        for (int wps = 1; wps <= 8; wps *= 2) {
            int blocks = cus * wps;   // 256 threads = 4 waves = 1 wave per SIMD per block
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, coef, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, coef, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, coef, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double per_wave = (double)iters * (kind == 1 ? 4 : 8);          // instructions per wave
            double instr = per_wave * wps;                                   // per SIMD
            double cycles = ms * 1e-3 * ghz * 1e9;
            double flops = (double)blocks * 256 * iters * 8 * (kind == 2 ? 1 : 2);
            printf("%-26s waves/SIMD %d: %.3f ms, %.3f wave-instr/cycle/SIMD, %.1f TFLOP/s\n", names[kind], wps, ms,
                   instr / cycles, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
