// What do the non-FMA instructions of the triangle filter cost on gfx950?  Each variant runs a
// loop of 64 vector instructions of a given mix at 4 waves per SIMD and prints cycles per
// wave-instruction per SIMD (nominal clock from argv[1], default 2.4 GHz):
//   0  v_pk_fma_f32                         4  56 v_pk_fma_f32 + 8 v_rcp_f32, one per 7
//   1  v_cmp_lt_f32 -> SGPR pair (VOP3)      5  56 v_pk_fma_f32 + 8 v_rcp_f32, the 8 in a row
//   2  v_cmp_lt_f32 -> VCC (VOPC)            6  48 v_pk_fma_f32 + 16 v_cmp (SGPR) + 8 s_or_b64
//   3  v_min3_f32                            7  v_rcp_f32 alone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int V>
__global__ void __launch_bounds__(256) k(float* out, const float* coef, int iters) {
    const float c0 = coef[0], c1 = coef[1];
    v2f a[8];
    float f[8];
    for (int i = 0; i < 8; i++) { a[i].x = threadIdx.x * 0.001f + i; a[i].y = a[i].x + 0.5f; f[i] = a[i].x; }
    const v2f k0 = {c0, c0}, k1 = {c1, c1};
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int n = rep * 8 + i;
                unsigned long long m;
                if (V == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k0), "v"(k1));
                if (V == 1) { asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(f[i]), "v"(c1)); acc |= m; }
                if (V == 2) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(f[i]), "v"(c1) : "vcc");
                if (V == 3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c0), "v"(c1));
                if (V == 4) {
                    if (n % 8 == 7) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(f[i]));
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k0), "v"(k1));
                }
                if (V == 5) {
                    if (rep == 7) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(f[i]));
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k0), "v"(k1));
                }
                if (V == 6) {
                    if (n % 4 == 3) { asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(f[i]), "v"(c1)); acc |= m; }
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k0), "v"(k1));
                }
                if (V == 7) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(f[i]));
                if (V == 8) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(f[i]), "v"(c1));
                if (V == 9) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n\ts_or_b64 %0, %0, vcc" : "+s"(acc) : "v"(f[i]), "v"(c1) : "vcc", "scc");
                if (V == 10) {      // three compares per "ray", combined the way the filter does
                    unsigned long long m1, m2, m3;
                    if (n % 4 == 3) {
                        asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(f[i]), "v"(c1));
                        asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(f[i]), "v"(c0));
                        asm volatile("v_cmp_le_f32_e64 %0, |%1|, %2" : "=s"(m3) : "v"(f[i]), "v"(c0));
                        acc |= (~(m1 | m2)) | m3;
                    } else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k0), "v"(k1));
                }
            }
        }
    }
    float s = (float)(acc & 0xffff);
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V> void run(int cus, float* out, float* coef, double ghz, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, wps = 4;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(cus * wps), dim3(256), 0, 0, out, coef, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr = (double)iters * 64 * wps, cycles = ms * 1e-3 * ghz * 1e9;
    printf("%d %-52s %.3f ms  %.2f cycles per vector instruction per SIMD\n", V, name, ms, cycles / instr);
}

int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out, *coef; hipMalloc(&out, (size_t)256 * cus * 8 * 4); hipMalloc(&coef, 8);
    float hc[2] = {0.999f, 0.001f}; hipMemcpy(coef, hc, 8, hipMemcpyHostToDevice);
    run<0>(cus, out, coef, ghz, "v_pk_fma_f32");
    run<1>(cus, out, coef, ghz, "v_cmp_lt_f32 -> SGPR pair, s_or_b64 each");
    run<2>(cus, out, coef, ghz, "v_cmp_lt_f32 -> VCC");
    run<3>(cus, out, coef, ghz, "v_min3_f32");
    run<4>(cus, out, coef, ghz, "56 v_pk_fma_f32 + 8 v_rcp_f32 spread");
    run<5>(cus, out, coef, ghz, "56 v_pk_fma_f32 + 8 v_rcp_f32 in a row");
    run<6>(cus, out, coef, ghz, "48 v_pk_fma_f32 + 16 v_cmp (SGPR) + 16 s_or_b64");
    run<7>(cus, out, coef, ghz, "v_rcp_f32");
    run<8>(cus, out, coef, ghz, "v_cmp_lt_f32 -> SGPR pair, result unused");
    run<9>(cus, out, coef, ghz, "v_cmp_lt_f32 -> VCC, s_or_b64 acc, acc, vcc");
    run<10>(cus, out, coef, ghz, "48 v_pk_fma_f32 + 16 x (3 v_cmp + SALU combine)");
    run<0>(cus, out, coef, ghz, "v_pk_fma_f32 (again, warm clock)");
    return 0;
}
