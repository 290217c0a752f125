// Isolated cycle costs of the scalar chain of k_propagate (one wave), via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../eqf_vio_amd/csrc/eqf_propagate.hpp"
using namespace eqf;
__global__ void k_bench(const Glob* gin, Glob* gout, PropArgs a, double* out) {
    const Glob& G = gin[0];
    ImuRec r = a.inl;
    int bad = 0;
    StepCommon c;
    long long t[8];
    t[0] = __builtin_readcyclecounter();
    stepCommon(G, r, a, c, kPartBase | kPartRicc | kPartLift, &bad);
    __asm__ volatile("" : "+v"(c.vC.x), "+v"(c.camInv.q.w), "+v"(c.Bg[0]), "+v"(c.Bvw.a[0]));
    t[1] = __builtin_readcyclecounter();
    const quat Qq = quat{0.9 + 1e-3 * threadIdx.x, 0.1, -0.2, 0.3};
    const LmBlocks blk = buildBlocks(c, Qq, 1.3, mk3(0.5, -0.2, 4.0 + threadIdx.x * 0.01));
    double acc = blk.D.a[0] + blk.Lw.a[3] + blk.Lv.a[7];
    __asm__ volatile("" : "+v"(acc));
    t[2] = __builtin_readcyclecounter();
    quat Qo; double ao;
    stepLandmark(c, a, Qq, 1.3, mk3(0.5, -0.2, 4.0 + threadIdx.x * 0.01), &Qo, &ao, &bad);
    acc += Qo.w + ao;
    __asm__ volatile("" : "+v"(acc));
    t[3] = __builtin_readcyclecounter();
    if (threadIdx.x == 48) stepGlobal(G, gout, r, a, c, &bad);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[4] = __builtin_readcyclecounter();
    double e0[3], cd[6], ci[6];
    poseConstants(quat{G.P0q[0], G.P0q[1], G.P0q[2], G.P0q[3]}, e0, cd, ci, &bad);
    acc += e0[0] + cd[1] + ci[2];
    __asm__ volatile("" : "+v"(acc));
    t[5] = __builtin_readcyclecounter();
    out[16 + threadIdx.x] = acc + bad;
    if (threadIdx.x == 0) for (int i = 0; i < 5; ++i) out[i] = double(t[i + 1] - t[i]);
}
int main() {
    Glob g; memset(&g, 0, sizeof(g));
    g.P0q[0] = 0.7; g.P0q[2] = 0.714; g.Aq[0] = 0.99; g.Aq[1] = 0.1; g.Aq[3] = 0.05; g.v0[0] = 0.3; g.curTime = 1.0; g.initialised = 1;
    g.curVel[0] = 0.1; g.curVel[1] = -0.2; g.curVel[2] = 0.05; g.curVel[3] = 9.7; g.curVel[5] = 0.5; g.eta0[0] = 1.0;
    g.cDiff[0] = 0.5; g.cDiff[4] = 0.5; g.cInv[0] = 2; g.cInv[3] = 2;
    Glob *gi, *go; hipMalloc(&gi, sizeof(Glob)); hipMalloc(&go, sizeof(Glob)); hipMemcpy(gi, &g, sizeof(Glob), hipMemcpyHostToDevice);
    PropArgs a; memset(&a, 0, sizeof(a));
    a.isImu = 1; a.doRiccati = 1; a.inl.stamp = 1.005; a.inl.w[0] = 0.1; a.inl.a[0] = 9.8;
    a.prm.useDiscreteVelocityLift = 1;
    for (int i = 0; i < 9; ++i) { a.prm.RIC[i] = a.prm.RICt[i] = a.prm.RcamI[i] = (i % 4 == 0); }
    a.prm.camq[0] = 1; a.prm.camIq[0] = 1;
    double* o; hipMalloc(&o, 4096);
    double h[5];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, gi, go, a, o);
        hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        printf("cycles: stepCommon %.0f  buildBlocks %.0f  stepLandmark %.0f  stepGlobal(1 lane) %.0f  poseConstants %.0f\n", h[0], h[1], h[2], h[3], h[4]);
    }
    return 0;
}
