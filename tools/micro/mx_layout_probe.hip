// Pins the operand conventions of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, OCP) by hypothesis testing on the host:
// random fp8 codes in every (lane, byte) of A and B and random run-time scales per lane go through ONE instruction; the host
// decodes the very register images it uploaded and evaluates C under each candidate convention.
//   pairing  P1: A(lane half, byte) meets B(same half, same byte)            (K order then does not matter)
//   scales   S1: a lane's scale applies to all 32 of its bytes  (block = lane half)
//            S2: bytes 0-15 of BOTH halves use the scale of the hh = 0 lane, bytes 16-31 that of the hh = 1 lane
//            S3: the reverse assignment of S2's halves within a lane (bytes 0-15 <- own half ... ) variants are listed below.
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/mx_layout_probe.bin tools/micro/mx_layout_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void one_mfma(const int* A, const int* B, const int* sa, const int* sb, float* C) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[lane * 8 + i]; b[i] = B[lane * 8 + i]; }
    f32x16 z;
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    const int s_a = sa[lane], s_b = sb[lane];               // run-time scale operands (byte 0)
    const f32x16 d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, z, 0, 0, 0, s_a, 0, s_b);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = d[i];
}

static double e4m3(int c) {                                  // OCP e4m3: bias 7, subnormals, no infinities
    const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
    const double v = e == 0 ? ldexp(m / 8.0, -6) : ldexp(1.0 + m / 8.0, e - 7);
    return s ? -v : v;
}

int main() {
    std::vector<int> A(64 * 8), B(64 * 8), sa(64), sb(64);
    srand(3);
    auto code = [] { int e = 5 + rand() % 5, m = rand() % 8, s = rand() % 2; return (s << 7) | (e << 3) | m; };   // |v| in [0.25, 7.5]
    for (int l = 0; l < 64; ++l) {
        for (int i = 0; i < 8; ++i) {
            A[l * 8 + i] = code() | (code() << 8) | (code() << 16) | (code() << 24);
            B[l * 8 + i] = code() | (code() << 8) | (code() << 16) | (code() << 24);
        }
        sa[l] = 124 + rand() % 7;
        sb[l] = 124 + rand() % 7;
    }
    int *dA, *dB, *dsa, *dsb; float* dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 4096);
    std::vector<float> C(1024);
    auto byte_of = [](const std::vector<int>& R, int lane, int e) { return (R[lane * 8 + e / 4] >> (8 * (e % 4))) & 255; };
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int> ua = sa, ub = sb;
        if (pass == 0) for (int l = 0; l < 64; ++l) ua[l] = ub[l] = 127;        // unit scales: the pairing alone
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, ua.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, ub.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        // hypotheses: (pairing, scale rule).  scale rule r: which lane's scale serves byte e of lane (row, hh)
        const char* names[] = {"S1: own lane for all 32 bytes", "S2: bytes 0-15 <- lane hh=0, bytes 16-31 <- lane hh=1",
                               "S3: bytes 0-15 <- lane hh=1, bytes 16-31 <- lane hh=0", "S4: always lane hh=0", "S5: always lane hh=1"};
        for (int pairing = 0; pairing < 2; ++pairing)
            for (int rule = 0; rule < (pass == 0 ? 1 : 5); ++rule) {
                double worst = 0;
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double acc = 0, nm = 0;
                        for (int hh = 0; hh < 2; ++hh)
                            for (int e = 0; e < 32; ++e) {
                                // pairing 0: identity; pairing 1: A(hh, e) meets B(hh ^ (e >= 16), e)  (a cross-half candidate)
                                const int hb = pairing == 0 ? hh : (hh ^ (e >= 16 ? 1 : 0));
                                int src = hh;
                                if (rule == 1) src = e < 16 ? 0 : 1; else if (rule == 2) src = e < 16 ? 1 : 0;
                                else if (rule == 3) src = 0; else if (rule == 4) src = 1;
                                const int srcb = (rule == 0) ? hb : src;
                                const double va = e4m3(byte_of(A, i + 32 * hh, e)) * ldexp(1.0, ua[i + 32 * src] - 127);
                                const double vb = e4m3(byte_of(B, j + 32 * hb, e)) * ldexp(1.0, ub[j + 32 * srcb] - 127);
                                acc += va * vb; nm += fabs(va * vb);
                            }
                        worst = fmax(worst, fabs(C[i * 32 + j] - acc) / nm);
                    }
                printf("%s | pairing %s | %s: max |C - model| / sum|ab| = %.2e %s\n", pass == 0 ? "unit scales  " : "random scales",
                       pairing == 0 ? "identity  " : "cross-half", pass == 0 ? "-" : names[rule], worst, worst < 1e-6 ? "  <== MATCH" : "");
            }
    }
    return 0;
}
