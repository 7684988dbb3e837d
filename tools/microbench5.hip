// Does the cost of v_pk_fma_f32 depend on which VGPR banks its 64-bit operands sit in?
// (bank = register number mod 4; an aligned pair covers banks {0,1} or {2,3}.)
// Explicit registers: 8 accumulator pairs and one multiplicand pair placed either in the same
// half of the banks as the accumulators or in the other half.  Cycles per instruction per SIMD
// at 4 waves/SIMD, nominal clock from argv[1].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// accumulators v[8:9], v[12:13], ... (banks {0,1}); multiplicand pair MUL
#define BODY(MUL)                                                            \
    "v_pk_fma_f32 v[8:9], s[4:5], " MUL ", v[8:9]\n"                         \
    "v_pk_fma_f32 v[12:13], s[4:5], " MUL ", v[12:13]\n"                     \
    "v_pk_fma_f32 v[16:17], s[4:5], " MUL ", v[16:17]\n"                     \
    "v_pk_fma_f32 v[20:21], s[4:5], " MUL ", v[20:21]\n"                     \
    "v_pk_fma_f32 v[24:25], s[4:5], " MUL ", v[24:25]\n"                     \
    "v_pk_fma_f32 v[28:29], s[4:5], " MUL ", v[28:29]\n"                     \
    "v_pk_fma_f32 v[32:33], s[4:5], " MUL ", v[32:33]\n"                     \
    "v_pk_fma_f32 v[36:37], s[4:5], " MUL ", v[36:37]\n"

#define CLOBBERS "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", "v36", "v37", \
                 "v40", "v41", "v42", "v43", "s4", "s5"

template <int V>
__global__ void __launch_bounds__(256) k(float* out, const float* coef, int iters) {
    float c0 = coef[0], c1 = coef[1];
    asm volatile("s_mov_b32 s4, %0\n s_mov_b32 s5, %0\n"
                 "v_mov_b32 v40, %1\n v_mov_b32 v41, %1\n v_mov_b32 v42, %1\n v_mov_b32 v43, %1\n"
                 "v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n"
                 "v_mov_b32 v16, %1\n v_mov_b32 v17, %1\n v_mov_b32 v20, %1\n v_mov_b32 v21, %1\n"
                 "v_mov_b32 v24, %1\n v_mov_b32 v25, %1\n v_mov_b32 v28, %1\n v_mov_b32 v29, %1\n"
                 "v_mov_b32 v32, %1\n v_mov_b32 v33, %1\n v_mov_b32 v36, %1\n v_mov_b32 v37, %1\n"
                 : : "s"(c0), "v"(c1) : CLOBBERS);
    for (int it = 0; it < iters; it++) {
        if (V == 0) asm volatile(BODY("v[40:41]") BODY("v[40:41]") BODY("v[40:41]") BODY("v[40:41]")
                                 BODY("v[40:41]") BODY("v[40:41]") BODY("v[40:41]") BODY("v[40:41]") : : : CLOBBERS);
        if (V == 1) asm volatile(BODY("v[42:43]") BODY("v[42:43]") BODY("v[42:43]") BODY("v[42:43]")
                                 BODY("v[42:43]") BODY("v[42:43]") BODY("v[42:43]") BODY("v[42:43]") : : : CLOBBERS);
    }
    float s;
    asm volatile("v_add_f32 %0, v8, v37" : "=v"(s) : : CLOBBERS);
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
    printf("%d %-64s %.3f ms  %.2f cycles per instruction per SIMD\n", V, name, ms, cycles / instr);
}

int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out, *coef; hipMalloc(&out, (size_t)256 * cus * 8 * 4); hipMalloc(&coef, 8);
    float hc[2] = {0.999f, 0.001f}; hipMemcpy(coef, hc, 8, hipMemcpyHostToDevice);
    run<0>(cus, out, coef, ghz, "warm-up");
    run<0>(cus, out, coef, ghz, "multiplicand pair in the accumulators' banks {0,1}");
    run<1>(cus, out, coef, ghz, "multiplicand pair in the other banks {2,3}");
    run<0>(cus, out, coef, ghz, "multiplicand pair in the accumulators' banks {0,1} (again)");
    return 0;
}
