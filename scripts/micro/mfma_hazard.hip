// How many wait states does gfx950 need between a v_mfma_f64_16x16x4_f64 and a VALU read of its result?  (LLVM's hazard recogniser inserts
// 18 for this DGEMM.)  A chain of CHAIN dependent MFMAs whose FIRST one moves the accumulator to fresh registers (vDst != SrcC), then NOPS
// wait states, then a v_mov of the result ("sample"); much later the same registers again ("final").  sample != final: the read was early.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NOPS, int CHAIN>
__global__ void k(const double* in, double* out) {
    const int lane = threadIdx.x;
    double a = in[lane], b = in[64 + lane], c0 = in[128 + lane], c1 = in[192 + lane], c2 = in[256 + lane], c3 = in[320 + lane], g = -7.0;
    double sample, fin;
    asm volatile(
        "v_mov_b64 v[100:101], %[a]\n\tv_mov_b64 v[102:103], %[b]\n\t"
        "v_mov_b64 v[104:105], %[c0]\n\tv_mov_b64 v[106:107], %[c1]\n\tv_mov_b64 v[108:109], %[c2]\n\tv_mov_b64 v[110:111], %[c3]\n\t"
        "v_mov_b64 v[112:113], %[g]\n\tv_mov_b64 v[114:115], %[g]\n\tv_mov_b64 v[116:117], %[g]\n\tv_mov_b64 v[118:119], %[g]\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        "v_mfma_f64_16x16x4_f64 v[112:119], v[100:101], v[102:103], v[104:111]\n\t"
        ".rept %c[chain]\n\tv_mfma_f64_16x16x4_f64 v[112:119], v[100:101], v[102:103], v[112:119]\n\t.endr\n\t"
        ".rept %c[nops]\n\ts_nop 0\n\t.endr\n\t"
        "v_mov_b64 %[s], v[112:113]\n\t"
        ".rept 16\n\ts_nop 15\n\t.endr\n\t"
        "v_mov_b64 %[f], v[112:113]\n\t"
        : [s] "=&v"(sample), [f] "=&v"(fin)
        : [a] "v"(a), [b] "v"(b), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [g] "v"(g), [nops] "n"(NOPS), [chain] "n"(CHAIN - 1)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
          "v117", "v118", "v119");
    out[lane] = sample;
    out[64 + lane] = fin;
}
template <int NOPS, int CHAIN>
void run(const double* din, double* dout) {
    double h[128];
    hipLaunchKernelGGL((k<NOPS, CHAIN>), dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    int early = 0, garbage = 0;
    for (int l = 0; l < 64; ++l) {
        if (h[l] != h[64 + l]) ++early;
        if (h[l] == -7.0) ++garbage;
    }
    printf("chain %2d, %3d wait states: %2d / 64 lanes read early (%2d still hold the pre-chain content)   sample[0] %.6g final[0] %.6g\n", CHAIN, NOPS, early,
        garbage, h[0], h[64]);
}
int main() {
    double hin[384];
    for (int i = 0; i < 384; ++i) hin[i] = 0.001 * (i % 97) + 0.5;
    double *din, *dout;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, 128 * 8);
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    run<0, 1>(din, dout); run<4, 1>(din, dout); run<8, 1>(din, dout); run<12, 1>(din, dout); run<16, 1>(din, dout); run<17, 1>(din, dout); run<18, 1>(din, dout);
    run<19, 1>(din, dout); run<20, 1>(din, dout); run<24, 1>(din, dout); run<32, 1>(din, dout); run<48, 1>(din, dout); run<64, 1>(din, dout); run<72, 1>(din, dout);
    run<0, 2>(din, dout); run<18, 2>(din, dout); run<32, 2>(din, dout); run<64, 2>(din, dout);
    run<0, 16>(din, dout); run<8, 16>(din, dout); run<18, 16>(din, dout); run<24, 16>(din, dout); run<32, 16>(din, dout); run<48, 16>(din, dout); run<64, 16>(din, dout);
    run<80, 16>(din, dout);
    return 0;
}
