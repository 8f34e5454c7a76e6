// What does v_mfma_f32_32x32x2_f32 do to the low bits?  C[32x32] = sum_k A[:,k] B[k,:] over K steps through a chain of MFMAs,
// compared bit for bit with host models of the accumulate: an fmaf chain (round to nearest even / toward zero), and
// "round each product, then add" chains.  Prints how many of the 1024 outputs each model reproduces and the mean signed
// error of the device result against a float64 sum (a non-zero mean = a biased rounding).
//   hipcc --offload-arch=gfx950 -O2 -frounding-math tools/mfma_rounding_probe.hip -o /tmp/mfma_rp && /tmp/mfma_rp
#include <hip/hip_runtime.h>
#include <cfenv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void chain(const float* A, const float* B, float* C, int K) {   // A [32][K], B [K][32]
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v16f acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) C[(8 * (e / 4) + 4 * h + e % 4) * 32 + r] = acc[e];
}

static float model(const float* a, const float* b, int K, int mode) {   // a: row of A, b: column of B (stride 32)
    float acc = 0.f;
    if (mode == 1 || mode == 3) fesetround(FE_TOWARDZERO); else fesetround(FE_TONEAREST);
    for (int k = 0; k < K; ++k) {
        if (mode <= 1) acc = fmaf(a[k], b[k * 32], acc);                                     // fma chain
        else { volatile float p = a[k] * b[k * 32]; acc = acc + p; }                         // product rounded, then added
    }
    fesetround(FE_TONEAREST);
    return acc;
}
static float model_pair(const float* a, const float* b, int K, bool rz) {                    // exact two-product sum per instruction, one rounding
    float acc = 0.f;
    for (int k = 0; k < K; k += 2) {
        const double t = (double)acc + (double)a[k] * b[k * 32] + (double)a[k + 1] * b[(k + 1) * 32];   // exact enough for a probe
        float f = (float)t;
        if (rz && fabs((double)f) > fabs(t)) f = nextafterf(f, 0.f);
        acc = f;
    }
    return acc;
}

int main() {
    for (int K : {2, 64, 2304}) {
        for (int positive = 0; positive < 2; ++positive) {
            std::vector<float> A(32 * K), B(K * 32), C(1024);
            srand(7);
            for (auto& v : A) v = (rand() / (float)RAND_MAX) * (positive ? 1.f : 2.f) - (positive ? 0.f : 1.f);
            for (auto& v : B) v = (rand() / (float)RAND_MAX) * (positive ? 1.f : 2.f) - (positive ? 0.f : 1.f);
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
            int hit[6] = {0, 0, 0, 0, 0, 0};
            double bias = 0, mag = 0, bias_rne = 0;
            for (int m = 0; m < 32; ++m)
                for (int n = 0; n < 32; ++n) {
                    const float* a = &A[m * K];
                    const float* b = &B[n];
                    const float c = C[m * 32 + n];
                    for (int mode = 0; mode < 4; ++mode) hit[mode] += model(a, b, K, mode) == c;
                    hit[4] += model_pair(a, b, K, false) == c;
                    hit[5] += model_pair(a, b, K, true) == c;
                    double ex = 0;
                    for (int k = 0; k < K; ++k) ex += (double)a[k] * b[k * 32];
                    bias += (c - ex) / (fabs(ex) + 1e-30);
                    bias_rne += (model(a, b, K, 0) - ex) / (fabs(ex) + 1e-30);
                    mag += fabs(ex);
                }
            printf("K=%4d %s operands: matches of 1024 -- fma chain RNE %4d, fma chain RZ %4d, mul+add RNE %4d, mul+add RZ %4d, pair-sum RNE %4d, pair-sum RZ %4d; "
                   "mean signed relative error of the device %.3e (host RNE fma chain %.3e)\n",
                   K, positive ? "positive" : "signed  ", hit[0], hit[1], hit[2], hit[3], hit[4], hit[5], bias / 1024, bias_rne / 1024);
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    }
    return 0;
}
