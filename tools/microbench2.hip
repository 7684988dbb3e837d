// Does scalar-ALU work steal issue bandwidth from the vector ALU on gfx950?
// 64 independent-ish v_fma_f32 per loop iteration, with S scalar instructions interleaved
// (S = 0, 16, 32, 64), at 4 and 8 waves per SIMD.  Prints VALU wave-instructions per cycle
// per SIMD at a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int S, int FORM>
__global__ void __launch_bounds__(256) k(float* out, const float* coef, int iters) {
    float c0 = coef[0], c1 = coef[1];
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    unsigned int sacc = 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (FORM == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c0), "v"(c1));
                if (FORM == 1) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "s"(c0), "v"(c1));
                if (FORM == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c1));
                if (FORM == 3) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c1));
                if (S >= 64 || (S == 32 && (i & 1) == 0) || (S == 16 && (i & 3) == 0))
                    asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) : : "scc");
            }
        }
    }
    float s = (float)sacc;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int S, int FORM> void run(int cus, float* out, float* coef, double ghz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps : {4, 8}) {
        int blocks = cus * wps;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<S, FORM>), dim3(blocks), dim3(256), 0, 0, out, coef, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        double valu = (double)iters * 64 * wps;
        double cycles = ms * 1e-3 * ghz * 1e9;
        printf("form %d SALU per 64 VALU = %2d, waves/SIMD %d: %.3f ms, VALU %.3f /cycle/SIMD, all instr %.3f /cycle/SIMD\n", FORM, S, wps,
               ms, valu / cycles, valu * (64.0 + S) / 64.0 / cycles);
    }
}

int main(int argc, char** argv) {
    double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    float *out, *coef; hipMalloc(&out, (size_t)256 * cus * 8 * 4); hipMalloc(&coef, 8);
    float hc[2] = {0.999f, 0.001f}; hipMemcpy(coef, hc, 8, hipMemcpyHostToDevice);
    run<0, 0>(cus, out, coef, ghz); run<0, 1>(cus, out, coef, ghz); run<0, 2>(cus, out, coef, ghz); run<0, 3>(cus, out, coef, ghz);
    run<16, 1>(cus, out, coef, ghz); run<32, 1>(cus, out, coef, ghz); run<64, 1>(cus, out, coef, ghz); run<32, 3>(cus, out, coef, ghz);
    return 0;
}
