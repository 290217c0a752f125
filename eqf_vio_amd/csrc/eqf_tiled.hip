// C ABI of the 2-D block-partitioned filter (include/eqf_vio_amd.h: "eqf_tiled_*"; include/eqf_vio_amd_debug.h: the dense tile kernels "eqf_tile_*"): the per-rank device side of
// BASELINE configs[4].  Second translation unit of libeqf_vio_amd.so; kernels in eqf_tiled.hpp (replicated O(N) state, base panel,
// local blocks) and eqf_tile.hpp (dense tile kernels of the distributed factorisations).  No CPU fallback: without a GPU
// eqf_tiled_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/eqf_vio_amd_debug.h"  // (the public header + the test / measurement hooks this library also exports)
#include "eqf_tile.hpp"
#include "eqf_tiled.hpp"

using namespace eqf;

#define HIPC(expr)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "eqf_vio_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return EQF_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

namespace {
// the caller's (torch's) current device is restored when an entry point returns
struct DeviceScope {
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) hipSetDevice(prev);
    }
};
// > 64 KB of dynamic LDS needs the function attribute once per device
int tileAttributes(int device) {
    static std::mutex mu;
    static std::vector<char> done;
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0) return EQF_ERR_INVALID;
    if ((int)done.size() <= device) done.resize(device + 1, 0);
    if (done[device]) return EQF_OK;
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_potrf), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_gemm_tn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLdsBytes));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_gemm_tn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLdsBytes));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_potrf_trail), hipFuncAttributeMaxDynamicSharedMemorySize, kTrailLdsBytes));
    done[device] = 1;
    return EQF_OK;
}
template <typename T>
int dmallocT(T** p, size_t count) {
    HIPC(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)));
    return EQF_OK;
}
}  // namespace

struct eqf_tiled {
    int cap = 0, device = 0, N = 0;
    eqf_settings set{};
    Params prm{};
    hipStream_t stream = nullptr;
    Glob* g[2] = {nullptr, nullptr};
    double* Q[2] = {nullptr, nullptr};
    double* Sb[2] = {nullptr, nullptr};
    int pG = 0, pB = 0, ldb = 0;
    double *p0 = nullptr, *lmc = nullptr, *blk = nullptr;
    CommonLds* blkCommon = nullptr;
    double *delta = nullptr, *Zrows = nullptr, *Vrows = nullptr, *Pg = nullptr, *Lgi = nullptr, *gamma = nullptr, *gammaTot = nullptr, *dBear = nullptr,
           *dOut = nullptr;
    int ldp = 0;
    int* errflag = nullptr;
    int *rowMap = nullptr, *colMap = nullptr;
    int nlr = 0, nlc = 0;
    // landmark slots (eqf_tiled.hpp): device flags, the host's copy of them, the marks of an edit
    int *active = nullptr, *mark = nullptr;
    std::vector<char> hostActive;
    // IMU bursts (eqf_tiled_propagate_burst): one record set, one base panel and one step-constant block per step of a burst, allocated at
    // the first burst
    double *histBlk = nullptr, *histBlkT = nullptr, *histSb = nullptr;
    // pinned staging of a frame's bearings (eqf_tiled_update_prep): the caller's array is copied before the call returns, the upload is
    // asynchronous -- an update is enqueued without waiting for the previous one
    double* hBear = nullptr;
    hipEvent_t evBear = nullptr;
    CommonLds* histCommon = nullptr;
    // host mirror of the control flow (VIOFilter.cpp:120-131, :146-152, :234-236)
    double curTime = -1.0;
    bool init = false;
};

namespace {
void freeTiled(eqf_tiled* t) {
    if (!t) return;
    for (int q = 0; q < 2; ++q) {
        hipFree(t->g[q]);
        hipFree(t->Q[q]);
        hipFree(t->Sb[q]);
    }
    for (void* p : {(void*)t->p0, (void*)t->lmc, (void*)t->blk, (void*)t->blkCommon, (void*)t->delta, (void*)t->Zrows, (void*)t->Vrows, (void*)t->Pg,
             (void*)t->Lgi, (void*)t->gamma, (void*)t->gammaTot, (void*)t->dBear, (void*)t->dOut, (void*)t->errflag, (void*)t->rowMap, (void*)t->colMap, (void*)t->active, (void*)t->mark, (void*)t->histBlk, (void*)t->histBlkT, (void*)t->histSb,
             (void*)t->histCommon})
        hipFree(p);
    if (t->hBear) hipHostFree(t->hBear);
    if (t->evBear) hipEventDestroy(t->evBear);
    delete t;
}
// camera-offset constants, same formulas as on the device (and as eqf_create)
void cameraConstants(Params& p) {
    const quat cq = quat{p.camq[0], p.camq[1], p.camq[2], p.camq[3]};
    const se3 camI = se3inv(se3{cq, mk3(p.camx[0], p.camx[1], p.camx[2])});
    const m33 RIC = q2m(cq), RICt = q2m(qinv(cq)), RcI = q2m(camI.q);
    for (int i = 0; i < 9; ++i) {
        p.RIC[i] = RIC.a[i];
        p.RICt[i] = RICt.a[i];
        p.RcamI[i] = RcI.a[i];
    }
    p.camIq[0] = camI.q.w; p.camIq[1] = camI.q.x; p.camIq[2] = camI.q.y; p.camIq[3] = camI.q.z;
    p.camIx[0] = camI.x.x; p.camIx[1] = camI.x.y; p.camIx[2] = camI.x.z;
}
int initTiledState(eqf_tiled* t) {
    Glob g0;
    std::memset(&g0, 0, sizeof(Glob));
    g0.P0q[0] = 1.0;
    g0.Aq[0] = 1.0;
    for (int i = 0; i < 3; ++i) {
        g0.bias[i] = t->set.initialOmegaBias[i];
        g0.bias[3 + i] = t->set.initialAccelBias[i];
    }
    g0.curTime = -1.0;
    std::vector<double> base((size_t)12 * t->ldb, 0.0);
    for (int i = 0; i < 3; ++i) {
        base[(size_t)i * t->ldb + i] = t->set.initialBiasOmegaVariance;
        base[(size_t)(3 + i) * t->ldb + 3 + i] = t->set.initialBiasAccelVariance;
        base[(size_t)(8 + i) * t->ldb + 8 + i] = t->set.initialVelocityVariance;
    }
    base[(size_t)6 * t->ldb + 6] = base[(size_t)7 * t->ldb + 7] = t->set.initialGravityVariance;
    for (int q = 0; q < 2; ++q) {
        HIPC(hipMemcpy(t->g[q], &g0, sizeof(Glob), hipMemcpyHostToDevice));
        HIPC(hipMemset(t->Q[q], 0, sizeof(double) * 5 * t->cap));
        HIPC(hipMemcpy(t->Sb[q], base.data(), sizeof(double) * 12 * t->ldb, hipMemcpyHostToDevice));
    }
    HIPC(hipMemset(t->p0, 0, sizeof(double) * 3 * t->cap));
    HIPC(hipMemset(t->lmc, 0, sizeof(double) * 15 * t->cap));  // (C = 0 for every slot that has never held a landmark)
    HIPC(hipMemset(t->errflag, 0, sizeof(int)));
    HIPC(hipMemset(t->active, 0, sizeof(int) * t->cap));
    t->hostActive.assign(t->cap, 0);
    t->pG = t->pB = 0;
    t->N = 0;
    t->curTime = -1.0;
    t->init = false;
    return EQF_OK;
}
TlArgs propArgs(eqf_tiled* t, const ImuRec& r, int isImu, double* Sll, int ldl) {
    TlArgs a{};
    a.gin = t->g[t->pG];
    a.gout = t->g[t->pG ^ 1];
    a.p0 = t->p0;
    a.Qin = t->Q[t->pG];
    a.Qout = t->Q[t->pG ^ 1];
    a.SbIn = t->Sb[t->pB];
    a.SbOut = t->Sb[t->pB ^ 1];
    a.ldb = t->ldb;
    a.cap = t->cap;
    a.inl = r;
    a.isImu = isImu;
    a.doRiccati = isImu ? (t->set.fastRiccati ? 0 : 1) : 1;  // VIOFilter.cpp:127, :233
    a.blk = t->blk;
    a.blkCommon = t->blkCommon;
    a.errflag = t->errflag;
    a.prm = t->prm;
    a.Sll = Sll;
    a.ldl = ldl;
    a.nlr = t->nlr;
    a.nlc = t->nlc;
    a.rowMap = t->rowMap;
    a.colMap = t->colMap;
    a.active = t->active;
    return a;
}
}  // namespace

extern "C" {

int eqf_tiled_create(const eqf_settings* settings, int capacity_landmarks, int device, eqf_tiled** out) {
    if (!settings || !out || capacity_landmarks < 1) return EQF_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return EQF_ERR_NO_DEVICE;
    DeviceScope ds(device);
    if (!ds.ok) return EQF_ERR_HIP;
    eqf_tiled* t = new eqf_tiled();
    t->cap = capacity_landmarks;
    t->device = device;
    t->set = *settings;
    Params& p = t->prm;
    p.biasOmegaProcessVariance = settings->biasOmegaProcessVariance;
    p.biasAccelProcessVariance = settings->biasAccelProcessVariance;
    p.gravityProcessVariance = settings->gravityProcessVariance;
    p.velocityProcessVariance = settings->velocityProcessVariance;
    p.pointProcessVariance = settings->pointProcessVariance;
    p.velOmegaVariance = settings->velOmegaVariance;
    p.velAccelVariance = settings->velAccelVariance;
    p.measurementVariance = settings->measurementVariance;
    p.initialPointVariance = settings->initialPointVariance;
    std::memcpy(p.camq, settings->cameraOffset_q, sizeof(p.camq));
    std::memcpy(p.camx, settings->cameraOffset_x, sizeof(p.camx));
    p.useInnovationLift = settings->useInnovationLift;
    p.useDiscreteInnovationLift = settings->useDiscreteInnovationLift;
    p.useDiscreteVelocityLift = settings->useDiscreteVelocityLift;
    cameraConstants(p);
    const int cap = t->cap;
    t->ldb = roundUp(kLm0 + 3 * cap, 16);
    t->ldp = roundUp(3 * cap, 16);
    int rc = EQF_OK;
    auto chk = [&](int r) { if (r && !rc) rc = r; };
    for (int q = 0; q < 2; ++q) {
        chk(dmallocT(&t->g[q], 1));
        chk(dmallocT(&t->Q[q], (size_t)5 * cap));
        chk(dmallocT(&t->Sb[q], (size_t)12 * t->ldb));
    }
    chk(dmallocT(&t->p0, (size_t)3 * cap));
    chk(dmallocT(&t->lmc, (size_t)15 * cap));
    chk(dmallocT(&t->blk, (size_t)kBlkRec * cap));
    chk(dmallocT(&t->blkCommon, 1));
    chk(dmallocT(&t->delta, (size_t)2 * cap));
    chk(dmallocT(&t->Zrows, (size_t)18 * cap));
    chk(dmallocT(&t->Vrows, (size_t)12 * cap));
    chk(dmallocT(&t->Pg, (size_t)5 * t->ldp));
    chk(dmallocT(&t->Lgi, 32));
    chk(dmallocT(&t->gamma, (size_t)kLm0 + 3 * cap));
    chk(dmallocT(&t->gammaTot, (size_t)9 + 3 * cap));
    chk(dmallocT(&t->dBear, (size_t)3 * cap));
    chk(dmallocT(&t->dOut, (size_t)16 + 3 * cap));
    chk(dmallocT(&t->errflag, 1));
    chk(dmallocT(&t->rowMap, cap));
    chk(dmallocT(&t->colMap, cap));
    chk(dmallocT(&t->active, cap));
    chk(dmallocT(&t->mark, cap));
    if (!rc) rc = initTiledState(t);
    if (!rc) rc = tileAttributes(device);
    if (rc) {
        freeTiled(t);
        return rc;
    }
    *out = t;
    return EQF_OK;
}

void eqf_tiled_destroy(eqf_tiled* t) {
    if (!t) return;
    DeviceScope ds(t->device);
    hipStreamSynchronize(t->stream);
    freeTiled(t);
}

int eqf_tiled_set_stream(eqf_tiled* t, void* stream) {
    if (!t) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    HIPC(hipStreamSynchronize(t->stream));
    t->stream = static_cast<hipStream_t>(stream);
    return EQF_OK;
}

int eqf_tiled_set_geometry(eqf_tiled* t, int nlr, const int* rowMap, int nlc, const int* colMap) {
    if (!t || nlr < 0 || nlc < 0 || nlr > t->cap || nlc > t->cap || (nlr && !rowMap) || (nlc && !colMap)) return EQF_ERR_INVALID;
    for (int i = 0; i < nlr; ++i)
        if (rowMap[i] < 0 || rowMap[i] >= t->cap) return EQF_ERR_INVALID;
    for (int i = 0; i < nlc; ++i)
        if (colMap[i] < 0 || colMap[i] >= t->cap) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    HIPC(hipStreamSynchronize(t->stream));
    if (nlr) HIPC(hipMemcpy(t->rowMap, rowMap, sizeof(int) * nlr, hipMemcpyHostToDevice));
    if (nlc) HIPC(hipMemcpy(t->colMap, colMap, sizeof(int) * nlc, hipMemcpyHostToDevice));
    t->nlr = nlr;
    t->nlc = nlc;
    return EQF_OK;
}

int eqf_tiled_propagate(eqf_tiled* t, double stamp, const double* omega, const double* accel, int is_imu, double* Sll, int ldl) {
    if (!t || (is_imu && (!omega || !accel))) return EQF_ERR_INVALID;
    if (t->N > 0 && t->nlr > 0 && t->nlc > 0 && (!Sll || ldl < 3 * t->nlc)) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    if (!ds.ok) return EQF_ERR_HIP;
    ImuRec r{};
    r.stamp = stamp;
    if (is_imu)
        for (int i = 0; i < 3; ++i) {
            r.w[i] = omega[i];
            r.a[i] = accel[i];
        }
    // host mirror of the control flow (VIOFilter.cpp:120-131, :146-152, :207, :234-236)
    int st = EQF_OK;
    if (t->curTime < 0) st = EQF_SKIPPED_BEFORE_FIRST_IMU;
    else if (!(stamp - t->curTime > 0)) st = EQF_SKIPPED_NONPOSITIVE_DT;
    const bool step = st == EQF_OK;
    const TlArgs a = propArgs(t, r, is_imu ? 1 : 0, Sll, ldl);
    const int N = t->N;
    hipLaunchKernelGGL(k_tl_build, dim3((std::max(N, 1) + 63) / 64 + 1), dim3(128), 0, t->stream, a);
    hipLaunchKernelGGL(k_tl_base, dim3(std::max(1, (N + 255) / 256)), dim3(256), 0, t->stream, a);
    if (step && a.doRiccati && N > 0 && t->nlr > 0 && t->nlc > 0)
        hipLaunchKernelGGL(k_tl_riccati, dim3((t->nlc + 255) / 256, (t->nlr + kStreamRows - 1) / kStreamRows), dim3(256), 0, t->stream, a);
    HIPC(hipGetLastError());
    t->pG ^= 1;
    t->pB ^= 1;
    if (is_imu) t->init = true;
    if (step || is_imu) t->curTime = stamp;
    if (!is_imu && st == EQF_OK && !t->init) st = EQF_SKIPPED_NOT_INITIALISED;
    return st;
}

int eqf_tiled_propagate_burst(eqf_tiled* t, int K, const double* stamps, const double* omega, const double* accel, int last_is_vision, double* Sll,
    int ldl, int* status) {
    if (!t || K < 1 || K > kTlBurstMax || !stamps || !status) return EQF_ERR_INVALID;
    const int nImu = K - (last_is_vision ? 1 : 0);
    if (nImu > 0 && (!omega || !accel)) return EQF_ERR_INVALID;
    const bool local = t->N > 0 && t->nlr > 0 && t->nlc > 0;
    if (local && (!Sll || ldl < 3 * t->nlc)) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    if (!ds.ok) return EQF_ERR_HIP;
    const int cap = t->cap;
    const size_t blkN = (size_t)kBlkRec * cap, blkTN = (size_t)27 * cap, sbN = (size_t)12 * t->ldb;
    if (!t->histBlk) {
        int rc = EQF_OK;
        auto chk = [&](int r) { if (r && !rc) rc = r; };
        chk(dmallocT(&t->histBlk, blkN * kTlBurstMax));
        chk(dmallocT(&t->histBlkT, blkTN * kTlBurstMax));
        chk(dmallocT(&t->histSb, sbN * kTlBurstMax));
        chk(dmallocT(&t->histCommon, kTlBurstMax));
        if (rc) return rc;
    }
    TlBurstArgs b{};
    const int N = t->N;
    for (int k = 0; k < K; ++k) {
        const bool isImu = k < nImu;
        ImuRec r{};
        r.stamp = stamps[k];
        if (isImu)
            for (int i = 0; i < 3; ++i) {
                r.w[i] = omega[3 * k + i];
                r.a[i] = accel[3 * k + i];
            }
        // host mirror of the control flow, call by call (VIOFilter.cpp:120-131, :146-152, :207, :234-236), as in eqf_tiled_propagate
        int st = EQF_OK;
        if (t->curTime < 0) st = EQF_SKIPPED_BEFORE_FIRST_IMU;
        else if (!(stamps[k] - t->curTime > 0)) st = EQF_SKIPPED_NONPOSITIVE_DT;
        const bool step = st == EQF_OK;
        TlArgs a = propArgs(t, r, isImu ? 1 : 0, Sll, ldl);
        // the burst's base panels: the current one, K - 1 in the history, the other half of the ping-pong pair
        a.SbIn = k == 0 ? t->Sb[t->pB] : t->histSb + sbN * k;
        a.SbOut = k == K - 1 ? t->Sb[t->pB ^ 1] : t->histSb + sbN * (k + 1);
        a.blk = t->histBlk + blkN * k;
        a.blkT = t->histBlkT + blkTN * k;
        a.blkCommon = t->histCommon + k;
        hipLaunchKernelGGL(k_tl_build, dim3((std::max(N, 1) + 63) / 64 + 1), dim3(128), 0, t->stream, a);
        hipLaunchKernelGGL(k_tl_base, dim3(std::max(1, (N + 255) / 256)), dim3(256), 0, t->stream, a);
        if (step && a.doRiccati && local) {
            const int q = b.nSteps++;
            b.blk[q] = a.blk;
            b.blkT[q] = a.blkT;
            b.SbIn[q] = a.SbIn;
            b.common[q] = a.blkCommon;
        }
        t->pG ^= 1;
        if (isImu) t->init = true;
        if (step || isImu) t->curTime = stamps[k];
        if (!isImu && st == EQF_OK && !t->init) st = EQF_SKIPPED_NOT_INITIALISED;
        status[k] = st;
    }
    t->pB ^= 1;
    if (b.nSteps > 0) {
        b.pointVar = t->prm.pointProcessVariance;
        b.ldb = t->ldb;
        b.cap = cap;
        b.Sll = Sll;
        b.ldl = ldl;
        b.nlr = t->nlr;
        b.nlc = t->nlc;
        b.rowMap = t->rowMap;
        b.colMap = t->colMap;
        hipLaunchKernelGGL(k_tl_riccati_burst, dim3((t->nlc + 255) / 256, (t->nlr + kBurstRows - 1) / kBurstRows), dim3(256), 0, t->stream, b);
    }
    HIPC(hipGetLastError());
    return EQF_OK;
}

int eqf_tiled_add_landmarks(eqf_tiled* t, int n, const double* bearings, double* Sll, int ldl) {
    if (!t || n < 1 || !bearings) return EQF_ERR_INVALID;
    if (n > t->cap) return EQF_ERR_CAPACITY;
    if (t->N != 0) return EQF_ERR_UNSUPPORTED;
    if (t->nlr > 0 && t->nlc > 0 && (!Sll || ldl < 3 * t->nlc)) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    if (!ds.ok) return EQF_ERR_HIP;
    HIPC(hipMemcpyAsync(t->dBear, bearings, sizeof(double) * 3 * n, hipMemcpyHostToDevice, t->stream));
    hipLaunchKernelGGL(k_tl_append, dim3((n + 127) / 128), dim3(128), 0, t->stream, t->g[0], t->g[1], n, t->set.initialSceneDepth, t->cap, t->dBear, t->p0,
        t->Q[0], t->Q[1], t->lmc, t->Sb[0], t->Sb[1], t->ldb, t->active, t->errflag);
    if (t->nlr > 0 && t->nlc > 0)
        hipLaunchKernelGGL(k_tl_init_local, dim3((t->nlc + 127) / 128, t->nlr), dim3(128), 0, t->stream, Sll, ldl, t->nlr, t->nlc, t->rowMap, t->colMap,
            t->set.initialPointVariance);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(t->stream));  // (bearings is pageable host memory)
    t->N = n;
    std::fill(t->hostActive.begin(), t->hostActive.begin() + n, 1);
    return EQF_OK;
}

int eqf_tiled_edit_landmarks(eqf_tiled* t, int n_remove, const int* remove_slots, int n_add, const int* add_slots, const double* add_bearings,
    double depth, int new_num_slots, double* Sll, int ldl) {
    if (!t || n_remove < 0 || n_add < 0 || (n_remove && !remove_slots) || (n_add && (!add_slots || !add_bearings))) return EQF_ERR_INVALID;
    if (new_num_slots < 0 || !(depth > 0.0) || !std::isfinite(depth)) return EQF_ERR_INVALID;
    if (new_num_slots > t->cap) return EQF_ERR_CAPACITY;
    const int n = std::max(t->N, new_num_slots);  // slots the edit may touch
    if (t->nlr > 0 && t->nlc > 0 && (!Sll || ldl < 3 * t->nlc)) return EQF_ERR_INVALID;
    // argument errors before any effect: every removed slot holds a landmark, every added slot is free (or freed by this call)
    std::vector<int> mark(t->cap, 0);  // (all of it is uploaded: the geometry in force may name any slot below the capacity)
    std::vector<double> bear((size_t)3 * std::max(n, 1), 0.0);
    for (int k = 0; k < n_remove; ++k) {
        const int s = remove_slots[k];
        if (s < 0 || s >= t->N || !t->hostActive[s] || mark[s]) return EQF_ERR_INVALID;
        mark[s] = 1;
    }
    for (int k = 0; k < n_add; ++k) {
        const int s = add_slots[k];
        if (s < 0 || s >= new_num_slots || mark[s] == 2 || (s < t->N && t->hostActive[s] && mark[s] != 1)) return EQF_ERR_INVALID;
        mark[s] = 2;
        for (int c = 0; c < 3; ++c) bear[(size_t)3 * s + c] = add_bearings[(size_t)3 * k + c];
    }
    for (int s = t->N; s < new_num_slots; ++s)  // the working set only grows over slots this call fills: a slot that was never initialised has
        if (mark[s] != 2) return EQF_ERR_INVALID;  // no identity constants / unit diagonal block (the hole invariants) to decouple it
    for (int s = new_num_slots; s < t->N; ++s)  // slots given up at the top must be empty after the edit
        if (t->hostActive[s] && mark[s] != 1) return EQF_ERR_INVALID;
    // the geometry must cover every slot in use before or after the edit: landmark blocks of marked slots are cleared through it
    DeviceScope ds(t->device);
    if (!ds.ok) return EQF_ERR_HIP;
    HIPC(hipMemcpyAsync(t->mark, mark.data(), sizeof(int) * t->cap, hipMemcpyHostToDevice, t->stream));
    if (n > 0) {
        HIPC(hipMemcpyAsync(t->dBear, bear.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice, t->stream));
    }
    hipLaunchKernelGGL(k_tl_edit_state, dim3((std::max(n, 1) + 127) / 128), dim3(128), 0, t->stream, t->g[0], t->g[1], n, new_num_slots, t->mark, depth,
        t->cap, t->dBear, t->p0, t->Q[0], t->Q[1], t->lmc, t->Sb[0], t->Sb[1], t->ldb, t->active, t->errflag);
    if (n > 0 && t->nlr > 0 && t->nlc > 0)
        hipLaunchKernelGGL(k_tl_edit_local, dim3((t->nlc + 127) / 128, t->nlr), dim3(128), 0, t->stream, Sll, ldl, t->nlr, t->nlc, t->rowMap, t->colMap,
            t->mark, t->set.initialPointVariance);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(t->stream));  // (mark / bear are pageable host memory)
    for (int s = 0; s < n; ++s)
        if (mark[s]) t->hostActive[s] = mark[s] == 2;
    t->N = new_num_slots;
    return EQF_OK;
}

// The host half of eqf_tiled_update_prep on its own: the frame's bearings (slot order) into the handle's pinned staging buffer.  A caller that
// replays a captured hipGraph of the update stages the bearings, then launches the graph (eqf_tiled_update_prep with bearings = NULL inside
// the capture: "already staged").
int eqf_tiled_stage_bearings(eqf_tiled* t, const double* bearings) {
    if (!t || !bearings || t->N < 1) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    if (!ds.ok) return EQF_ERR_HIP;
    if (!t->hBear) {
        HIPC(hipHostMalloc(reinterpret_cast<void**>(&t->hBear), sizeof(double) * 3 * t->cap, hipHostMallocDefault));
        HIPC(hipEventCreateWithFlags(&t->evBear, hipEventDisableTiming));
    }
    HIPC(hipEventSynchronize(t->evBear));  // (the previous frame's upload has left the staging buffer: it did long ago)
    std::memcpy(t->hBear, bearings, sizeof(double) * 3 * t->N);
    return EQF_OK;
}
// (library-internal, csrc/eqf_tiledf.hip) A captured hipGraph of an update contains the upload of the staged bearings, but an event recorded
// during capture is a node of the graph, not a record a host can wait on: after every replay the host loop records evBear here, on the
// stream the graph was launched on, so that eqf_tiled_stage_bearings waits for THIS frame's copy node before it overwrites the buffer.
extern "C" __attribute__((visibility("hidden"))) int eqf_tiled_bearings_consumed(eqf_tiled* t, void* stream) {
    if (!t || !t->evBear) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    if (!ds.ok) return EQF_ERR_HIP;
    HIPC(hipEventRecord(t->evBear, static_cast<hipStream_t>(stream)));
    return EQF_OK;
}
// bit 0: which of the two scalar-state / landmark buffers is current, bit 1: which base panel (they alternate with every step / burst: the
// device pointers an update's launches carry depend on them)
int eqf_tiled_pingpong(eqf_tiled* t) { return t ? (t->pG | (t->pB << 1)) : EQF_ERR_INVALID; }

int eqf_tiled_update_prep(eqf_tiled* t, const double* bearings, const double* Sll, int ldl, double* M, int ldm, double* E, int lde, double* G11) {
    if (!t || !G11 || t->N < 1 || (!bearings && !t->hBear)) return EQF_ERR_INVALID;
    const bool local = t->nlr > 0 && t->nlc > 0;
    if (local && (!Sll || !M || !E || ldl < 3 * t->nlc || ldm < 5 * t->nlc + kTlNarrowS || lde < 3 * t->nlc + kTlNarrowE)) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    if (!ds.ok) return EQF_ERR_HIP;
    const int N = t->N;
    if (bearings) {
        int rcs = eqf_tiled_stage_bearings(t, bearings);
        if (rcs) return rcs;
    }
    HIPC(hipMemcpyAsync(t->dBear, t->hBear, sizeof(double) * 3 * N, hipMemcpyHostToDevice, t->stream));
    HIPC(hipEventRecord(t->evBear, t->stream));
    TlUpdArgs a{};
    a.g = t->g[t->pG];
    a.p0 = t->p0;
    a.lmc = t->lmc;
    a.Q = t->Q[t->pG];
    a.Sb = t->Sb[t->pB];
    a.ldb = t->ldb;
    a.cap = t->cap;
    a.bearings = t->dBear;
    a.delta = t->delta;
    a.Zrows = t->Zrows;
    a.Vrows = t->Vrows;
    a.Pg = t->Pg;
    a.Lgi = t->Lgi;
    a.ldp = t->ldp;
    a.errflag = t->errflag;
    a.prm = t->prm;
    a.Sll = Sll;
    a.ldl = ldl;
    a.nlr = t->nlr;
    a.nlc = t->nlc;
    a.rowMap = t->rowMap;
    a.colMap = t->colMap;
    a.active = t->active;
    a.M = M;
    a.ldm = ldm;
    a.E = E;
    a.lde = lde;
    a.G11 = G11;
    hipLaunchKernelGGL(k_tl_prep, dim3((N + 63) / 64), dim3(64), 0, t->stream, a);
    hipLaunchKernelGGL(k_tl_eprep, dim3((3 * N + 255) / 256), dim3(256), 0, t->stream, a);
    if (local) {
        const dim3 grid((t->nlc + 255) / 256 + 1, (t->nlr + kStreamRows - 1) / kStreamRows);
        hipLaunchKernelGGL(k_tl_form_s, grid, dim3(256), 0, t->stream, a);
        hipLaunchKernelGGL(k_tl_form_e, grid, dim3(256), 0, t->stream, a);
    }
    HIPC(hipGetLastError());
    return EQF_OK;
}

int eqf_tiled_update_finish(eqf_tiled* t, const double* acc, int ldacc, const double* Gnn, const double* G11) {
    if (!t || !acc || !Gnn || !G11 || t->N < 1 || ldacc < 3 * t->N) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    if (!ds.ok) return EQF_ERR_HIP;
    TlFinArgs a{};
    a.u.g = t->g[t->pG];
    a.u.p0 = t->p0;
    a.u.lmc = t->lmc;
    a.u.Q = t->Q[t->pG];
    a.u.cap = t->cap;
    a.u.dbgDelta = t->delta;
    a.u.dbgGamma = t->gamma;
    a.u.dbgGammaTot = t->gammaTot;
    a.u.errflag = t->errflag;
    a.u.prm = t->prm;
    a.acc = acc;
    a.ldacc = ldacc;
    a.Gnn = Gnn;
    a.G11 = G11;
    a.Sb = t->Sb[t->pB];
    a.ldb = t->ldb;
    hipLaunchKernelGGL(k_tl_finish, dim3(1), dim3(256), 0, t->stream, a);
    HIPC(hipGetLastError());
    return EQF_OK;
}

int eqf_tiled_synchronize(eqf_tiled* t) {
    if (!t) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    HIPC(hipStreamSynchronize(t->stream));
    return EQF_OK;
}
int eqf_tiled_num_landmarks(eqf_tiled* t) { return t ? t->N : EQF_ERR_INVALID; }
int eqf_tiled_get_time(eqf_tiled* t, double* time) {
    if (!t || !time) return EQF_ERR_INVALID;
    *time = t->curTime;
    return EQF_OK;
}
int eqf_tiled_device_error(eqf_tiled* t) {
    if (!t) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    HIPC(hipStreamSynchronize(t->stream));
    int e = 0;
    HIPC(hipMemcpy(&e, t->errflag, sizeof(int), hipMemcpyDeviceToHost));
    return e;
}

static int fetchGlob(eqf_tiled* t, Glob* g) {
    HIPC(hipStreamSynchronize(t->stream));
    HIPC(hipMemcpy(g, t->g[t->pG], sizeof(Glob), hipMemcpyDeviceToHost));
    return EQF_OK;
}

int eqf_tiled_get_state_estimate(eqf_tiled* t, double* pose_q, double* pose_x, double* velocity, double* p) {
    if (!t) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    const int N = t->N;
    hipLaunchKernelGGL(k_tl_state_estimate, dim3(std::max(1, (N + 127) / 128)), dim3(128), 0, t->stream, t->g[t->pG], t->p0, t->Q[t->pG], t->cap, t->dOut);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(t->stream));
    std::vector<double> h((size_t)10 + 3 * N);
    HIPC(hipMemcpy(h.data(), t->dOut, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    if (pose_q) std::copy(h.begin(), h.begin() + 4, pose_q);
    if (pose_x) std::copy(h.begin() + 4, h.begin() + 7, pose_x);
    if (velocity) std::copy(h.begin() + 7, h.begin() + 10, velocity);
    if (p) std::copy(h.begin() + 10, h.end(), p);
    return EQF_OK;
}

static int fetchSoA(eqf_tiled* t, const double* src, int rows, double* dst /* [N][rows] */) {
    const int N = t->N, cap = t->cap;
    std::vector<double> h((size_t)rows * cap);
    HIPC(hipMemcpy(h.data(), src, sizeof(double) * rows * cap, hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i)
        for (int c = 0; c < rows; ++c) dst[(size_t)i * rows + c] = h[(size_t)c * cap + i];
    return EQF_OK;
}

int eqf_tiled_get_origin(eqf_tiled* t, double* pose_q, double* pose_x, double* velocity, double* p) {
    if (!t) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    Glob g;
    int rc = fetchGlob(t, &g);
    if (rc) return rc;
    if (pose_q) std::copy(g.P0q, g.P0q + 4, pose_q);
    if (pose_x) std::copy(g.P0x, g.P0x + 3, pose_x);
    if (velocity) std::copy(g.v0, g.v0 + 3, velocity);
    if (p) return fetchSoA(t, t->p0, 3, p);
    return EQF_OK;
}

int eqf_tiled_get_group(eqf_tiled* t, double* A_q, double* A_x, double* w, double* Q_q, double* Q_a) {
    if (!t) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    Glob g;
    int rc = fetchGlob(t, &g);
    if (rc) return rc;
    if (A_q) std::copy(g.Aq, g.Aq + 4, A_q);
    if (A_x) std::copy(g.Ax, g.Ax + 3, A_x);
    if (w) std::copy(g.w, g.w + 3, w);
    if (Q_q || Q_a) {
        std::vector<double> q5((size_t)5 * std::max(t->N, 1));
        rc = fetchSoA(t, t->Q[t->pG], 5, q5.data());
        if (rc) return rc;
        for (int i = 0; i < t->N; ++i) {
            if (Q_q) std::copy(q5.begin() + 5 * i, q5.begin() + 5 * i + 4, Q_q + 4 * i);
            if (Q_a) Q_a[i] = q5[5 * i + 4];
        }
    }
    return EQF_OK;
}

int eqf_tiled_get_bias(eqf_tiled* t, double* bias6) {
    if (!t || !bias6) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    Glob g;
    int rc = fetchGlob(t, &g);
    if (rc) return rc;
    std::copy(g.bias, g.bias + 6, bias6);
    return EQF_OK;
}

int eqf_tiled_get_integrator(eqf_tiled* t, double* currentVelocity6, double* accumulatedVelocity6, double* accumulatedTime, int* initialised) {
    if (!t) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    Glob g;
    int rc = fetchGlob(t, &g);
    if (rc) return rc;
    if (currentVelocity6) std::copy(g.curVel, g.curVel + 6, currentVelocity6);
    if (accumulatedVelocity6) std::copy(g.accVel, g.accVel + 6, accumulatedVelocity6);
    if (accumulatedTime) *accumulatedTime = g.accTime;
    if (initialised) *initialised = g.initialised;
    return EQF_OK;
}

int eqf_tiled_get_last_update(eqf_tiled* t, double* delta, double* gamma, double* Gamma) {
    if (!t) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    HIPC(hipStreamSynchronize(t->stream));
    const int N = t->N;
    if (delta) HIPC(hipMemcpy(delta, t->delta, sizeof(double) * 2 * N, hipMemcpyDeviceToHost));
    if (gamma) {
        std::vector<double> h((size_t)kLm0 + 3 * N);
        HIPC(hipMemcpy(h.data(), t->gamma, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
        std::copy(h.begin(), h.begin() + 11, gamma);
        std::copy(h.begin() + kLm0, h.end(), gamma + 11);
    }
    if (Gamma) HIPC(hipMemcpy(Gamma, t->gammaTot, sizeof(double) * (9 + 3 * N), hipMemcpyDeviceToHost));
    return EQF_OK;
}

int eqf_tiled_get_base(eqf_tiled* t, double* dst, int ld) {
    if (!t || !dst || ld < 11 + 3 * t->N) return EQF_ERR_INVALID;
    DeviceScope ds(t->device);
    HIPC(hipStreamSynchronize(t->stream));
    const int N = t->N;
    std::vector<double> h((size_t)12 * t->ldb);
    HIPC(hipMemcpy(h.data(), t->Sb[t->pB], sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < 11; ++r) {
        for (int c = 0; c < 11; ++c) dst[(size_t)r * ld + c] = h[(size_t)r * t->ldb + c];
        for (int c = 0; c < 3 * N; ++c) dst[(size_t)r * ld + 11 + c] = h[(size_t)r * t->ldb + kLm0 + c];
    }
    return EQF_OK;
}

int eqf_tiled_set_state(eqf_tiled* t, int N, const double* pose_q, const double* pose_x, const double* velocity, const double* p0,
    const double* A_q, const double* A_x, const double* w, const double* Q_q, const double* Q_a, const double* bias6, const double* sigma_base,
    int ld, double currentTime, const double* currentVelocity6, const double* accumulatedVelocity6, double accumulatedTime, int initialised) {
    if (!t || N < 0 || !pose_q || !pose_x || !velocity || !A_q || !A_x || !w || !bias6 || !sigma_base || ld < 11 + 3 * N) return EQF_ERR_INVALID;
    if (N > 0 && (!p0 || !Q_q || !Q_a)) return EQF_ERR_INVALID;
    if (N > t->cap) return EQF_ERR_CAPACITY;
    DeviceScope ds(t->device);
    HIPC(hipStreamSynchronize(t->stream));
    const int cap = t->cap;
    Glob g;
    std::memset(&g, 0, sizeof(Glob));
    std::copy(pose_q, pose_q + 4, g.P0q);
    std::copy(pose_x, pose_x + 3, g.P0x);
    std::copy(velocity, velocity + 3, g.v0);
    std::copy(A_q, A_q + 4, g.Aq);
    std::copy(A_x, A_x + 3, g.Ax);
    std::copy(w, w + 3, g.w);
    std::copy(bias6, bias6 + 6, g.bias);
    if (currentVelocity6) std::copy(currentVelocity6, currentVelocity6 + 6, g.curVel);
    if (accumulatedVelocity6) std::copy(accumulatedVelocity6, accumulatedVelocity6 + 6, g.accVel);
    g.accTime = accumulatedTime;
    g.curTime = currentTime;
    g.initialised = initialised ? 1 : 0;
    g.N = N;
    std::vector<double> hp((size_t)3 * cap, 0.0), hq((size_t)5 * cap, 0.0), hb((size_t)12 * t->ldb, 0.0);
    for (int i = 0; i < N; ++i) {
        for (int c = 0; c < 3; ++c) hp[(size_t)c * cap + i] = p0[3 * i + c];
        for (int c = 0; c < 4; ++c) hq[(size_t)c * cap + i] = Q_q[4 * i + c];
        hq[(size_t)4 * cap + i] = Q_a[i];
    }
    for (int r = 0; r < 11; ++r) {
        for (int c = 0; c < 11; ++c) hb[(size_t)r * t->ldb + c] = sigma_base[(size_t)r * ld + c];
        for (int c = 0; c < 3 * N; ++c) hb[(size_t)r * t->ldb + kLm0 + c] = sigma_base[(size_t)r * ld + 11 + c];
    }
    for (int q = 0; q < 2; ++q) {
        HIPC(hipMemcpy(t->g[q], &g, sizeof(Glob), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(t->Q[q], hq.data(), sizeof(double) * hq.size(), hipMemcpyHostToDevice));
        HIPC(hipMemcpy(t->Sb[q], hb.data(), sizeof(double) * hb.size(), hipMemcpyHostToDevice));
    }
    HIPC(hipMemcpy(t->p0, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice));
    HIPC(hipMemset(t->active, 0, sizeof(int) * cap));
    for (int q = 0; q < 2; ++q) {
        hipLaunchKernelGGL(k_tl_restore, dim3(std::max(1, (N + 127) / 128)), dim3(128), 0, t->stream, t->g[q], t->p0, t->lmc, cap, t->active, t->errflag);
    }
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(t->stream));
    t->hostActive.assign(cap, 0);
    std::fill(t->hostActive.begin(), t->hostActive.begin() + N, 1);
    t->N = N;
    t->curTime = currentTime;
    t->init = initialised != 0;
    return EQF_OK;
}

// ---- streams restricted to a set of CUs ---------------------------------------------------------------------------------
// A CU-masked stream is a hardware queue of its own, and the queues of destroyed streams were measured NOT to come back (round 6: the
// seventh partitioned filter created and destroyed in one process ran at 114 ms a frame instead of 65 -- seven streams a handle, the
// queues oversubscribed).  So a masked stream that is handed back is kept, per device and mask, and the next request for that mask gets
// it: a process that creates and destroys handles for hours holds as many queues as its live handles ever needed at once.
extern "C++" {
namespace {
struct MaskedPool {
    std::mutex mu;
    std::map<std::pair<int, std::vector<uint32_t>>, std::vector<hipStream_t>> idle;
    std::map<hipStream_t, std::pair<int, std::vector<uint32_t>>> made;
};
MaskedPool& maskedPool();
// at process exit, before the runtime's own teardown (registered after it was initialised, so run before it): the kept streams go.  (Left alive
// they crashed rocprofv3's finaliser at exit -- rc 139 after the trace had been written.)
void drainMaskedPool() {
    MaskedPool& pool = maskedPool();
    std::lock_guard<std::mutex> lk(pool.mu);
    for (auto& kv : pool.idle)
        for (hipStream_t st : kv.second) {
            pool.made.erase(st);
            (void)hipStreamDestroy(st);
        }
    pool.idle.clear();
}
MaskedPool& maskedPool() {
    // (the pool object itself is never destroyed: handles may be closed from destructors that run at process exit)
    static MaskedPool* p = [] {
        MaskedPool* q = new MaskedPool;
        std::atexit(drainMaskedPool);
        return q;
    }();
    return *p;
}
}  // namespace
}  // extern "C++"
int eqf_stream_create_masked(int device, int first_cu, int num_cus, int complement, void** out) {
    if (!out || first_cu < 0 || num_cus < 1) return EQF_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return EQF_ERR_NO_DEVICE;
    DeviceScope ds(device);
    if (!ds.ok) return EQF_ERR_HIP;
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount;
    if (first_cu + num_cus > cus || (complement && num_cus >= cus)) return EQF_ERR_INVALID;
    std::vector<uint32_t> mask((cus + 31) / 32, 0u);
    for (int c = 0; c < cus; ++c) {
        const bool in = c >= first_cu && c < first_cu + num_cus;
        if (in != (complement != 0)) mask[c / 32] |= 1u << (c % 32);
    }
    MaskedPool& pool = maskedPool();
    const auto key = std::make_pair(device, mask);
    std::lock_guard<std::mutex> lk(pool.mu);
    auto it = pool.idle.find(key);
    if (it != pool.idle.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        return EQF_OK;
    }
    hipStream_t st = nullptr;
    HIPC(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    pool.made[st] = key;
    *out = st;
    return EQF_OK;
}
int eqf_stream_destroy(int device, void* stream) {
    if (!stream) return EQF_OK;
    DeviceScope ds(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPC(hipStreamSynchronize(st));
    MaskedPool& pool = maskedPool();
    static const bool keep = !(std::getenv("EQF_STREAM_POOL") && std::atoi(std::getenv("EQF_STREAM_POOL")) == 0);  // (EQF_STREAM_POOL=0: destroy)
    if (keep) {
        std::lock_guard<std::mutex> lk(pool.mu);
        auto it = pool.made.find(st);
        if (it != pool.made.end()) {  // (idle, in order: whatever was queued on it has finished)
            pool.idle[it->second].push_back(st);
            return EQF_OK;
        }
    }
    HIPC(hipStreamDestroy(st));
    return EQF_OK;
}

// ---- dense tile kernels ------------------------------------------------------------------------------------------------------
int eqf_tile_gemm_tn(int device, void* stream, double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k,
    double alpha, int mask_rb, int mask_cb, int rblk0, int Pr, int pr, int cblk0, int Pc, int pc) {
    if (!C || !A || !B || m < 1 || n < 1 || k < 1 || ldc < n || lda < m || ldb < n) return EQF_ERR_INVALID;
    if (mask_rb < 0 || mask_cb < 0 || ((mask_rb > 0) != (mask_cb > 0)) || (mask_rb > 0 && (Pr < 1 || Pc < 1))) return EQF_ERR_INVALID;
    DeviceScope ds(device);
    if (!ds.ok) return EQF_ERR_HIP;
    int rc = tileAttributes(device);
    if (rc) return rc;
    GemmMask mk{mask_rb, mask_cb, rblk0, Pr, pr, cblk0, Pc, pc};
    const int tm = (m + kGemmTile - 1) / kGemmTile, tn = (n + kGemmTile - 1) / kGemmTile;
    GemmPlan pl;
    gemmMakePlan(mk, tm, tn, n, &pl);  // (masked launches: only the tiles the mask keeps are dispatched, XCD-contiguous like the others)
    if (mask_rb > 0 && tm <= kGemmPlanRows && pl.tact == 0) return EQF_OK;  // the mask leaves nothing
    const long long T = pl.tact > 0 ? pl.tact : (long long)tm * tn;
    const int grid = int(8 * ((T + 7) / 8));
    // the DIRECT build (operand rows global -> LDS without a register stop) unless EQF_GEMM_DIRECT=0: same arithmetic in the same order, bitwise
    // the same C; 55.4 -> 57.6 / 58.3 -> 60.3 / 51.8 -> 59.1 TFLOP/s at the update's shapes, cfg 5 64.6 -> 63.3 ms a frame (profiles/r06_gemm_direct.txt)
    static const bool direct = !(std::getenv("EQF_GEMM_DIRECT") && std::atoi(std::getenv("EQF_GEMM_DIRECT")) == 0);
    if (direct)
        hipLaunchKernelGGL(k_tile_gemm_tn<true>, dim3(grid), dim3(256), kGemmLdsBytes, static_cast<hipStream_t>(stream), C, ldc, m, n, A, lda, B, ldb, k, alpha,
            mk, tm, tn, pl);
    else
        hipLaunchKernelGGL(k_tile_gemm_tn<false>, dim3(grid), dim3(256), kGemmLdsBytes, static_cast<hipStream_t>(stream), C, ldc, m, n, A, lda, B, ldb, k, alpha,
            mk, tm, tn, pl);
    HIPC(hipGetLastError());
    return EQF_OK;
}

int eqf_tile_mirror(int device, void* stream, double* C, int ldc, int n, int rb) {
    if (!C || n < 1 || ldc < n || rb < 1) return EQF_ERR_INVALID;
    DeviceScope ds(device);
    if (!ds.ok) return EQF_ERR_HIP;
    hipLaunchKernelGGL(k_tile_mirror, dim3((n + 63) / 64, (n + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), C, ldc, n, rb);
    HIPC(hipGetLastError());
    return EQF_OK;
}

int eqf_tile_downdate(int device, void* stream, double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k) {
    return eqf_tile_gemm_tn(device, stream, C, ldc, m, n, A, lda, B, ldb, k, -1.0, 0, 0, 0, 1, 0, 0, 1, 0);
}

// ---- products on the integer matrix pipe (csrc/eqf_tile.hpp: k_i8_colexp / k_i8_split / k_i8_gemm): the downdate, the factorisations' trailing updates
extern "C++" {
namespace {
// aOff >= 0: A is the columns [aOff, aOff + m) of B (aOff a multiple of 32): one split serves both sides, A's tiles are B's from tile aOff / 32 on
struct I8Plan {
    int mp, np, ntB, nKc;
    size_t sliceA, sliceB, offB, offEA, offEB, total;
};
I8Plan i8Plan(int m, int n, int k, int slices, int aOff) {
    I8Plan p;
    p.mp = (m + 127) / 128 * 128;  // rows of C: 128 per workgroup
    p.np = (n + 63) / 64 * 64;     // columns of C: 64 per workgroup
    p.ntB = aOff >= 0 ? std::max(p.np, aOff + p.mp) : p.np;  // columns of B that are cut (past n: zero)
    p.nKc = (k + 31) / 32;
    p.sliceA = aOff >= 0 ? 0 : (size_t)(p.mp / 32) * p.nKc * slices * 1024;
    p.sliceB = (size_t)(p.ntB / 32) * p.nKc * slices * 1024;
    p.offB = p.sliceA;
    p.offEA = p.sliceA + p.sliceB;
    p.offEB = p.offEA + (aOff >= 0 ? 0 : sizeof(int) * (size_t)p.mp);
    p.total = p.offEB + sizeof(int) * (size_t)p.ntB;
    return p;
}
template <int S>
int i8Product(hipStream_t st, double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k, const GemmMask& mk,
    int maskCols, int aOff, char* ws, const I8Plan& p) {
    signed char* sB = reinterpret_cast<signed char*>(ws + p.offB);
    int* eB = reinterpret_cast<int*>(ws + p.offEB);
    signed char* sA = aOff >= 0 ? sB + (size_t)(aOff / 32) * p.nKc * S * 1024 : reinterpret_cast<signed char*>(ws);
    int* eA = aOff >= 0 ? eB + aOff : reinterpret_cast<int*>(ws + p.offEA);
    const int nExp = (int)((p.total - p.offEA) / sizeof(int));  // (a kernel, not hipMemsetAsync: 32 of these per update on CU-masked streams)
    hipLaunchKernelGGL(k_i8_zero, dim3((nExp + 255) / 256), dim3(256), 0, st, reinterpret_cast<int*>(ws + p.offEA), nExp);
    const int kslabs = (k + 511) / 512;
    if (aOff < 0) {
        hipLaunchKernelGGL(k_i8_colexp, dim3((m + 63) / 64, kslabs), dim3(256), 0, st, A, k, m, lda, eA);
        hipLaunchKernelGGL(k_i8_split<S>, dim3(p.mp / 32, (p.nKc + 3) / 4), dim3(256), 0, st, A, k, m, lda, eA, sA, p.nKc);
    }
    hipLaunchKernelGGL(k_i8_colexp, dim3((n + 63) / 64, kslabs), dim3(256), 0, st, B, k, n, ldb, eB);
    hipLaunchKernelGGL(k_i8_split<S>, dim3(p.ntB / 32, (p.nKc + 3) / 4), dim3(256), 0, st, B, k, n, ldb, eB, sB, p.nKc);
    hipLaunchKernelGGL(k_i8_gemm<S>, dim3(p.np / 64, p.mp / 128), dim3(512), 0, st, sA, sB, eA, eB, C, m, n, ldc, p.nKc, -1.0, mk, maskCols);
    HIPC(hipGetLastError());
    return EQF_OK;
}
}  // namespace
}  // extern "C++"

size_t eqf_tile_i8_workspace_bytes(int m, int n, int k, int slices, int same_operand) {
    if (m < 1 || n < 1 || k < 1 || slices < 5 || slices > 7) return 0;
    return i8Plan(m, n, k, slices, same_operand && m == n ? 0 : -1).total;
}

int eqf_tile_gemm_tn_i8(int device, void* stream, double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k,
    int slices, int mask_rb, int mask_cb, int rblk0, int Pr, int pr, int cblk0, int Pc, int pc, int mask_cols, void* workspace,
    size_t workspace_bytes) {
    if (!C || !A || !B || !workspace || m < 1 || n < 1 || k < 1 || k > 70000 || ldc < n || lda < m || ldb < n) return EQF_ERR_INVALID;
    if (slices < 5 || slices > 7) return EQF_ERR_INVALID;
    if (mask_rb < 0 || mask_cb < 0 || (mask_rb > 0) != (mask_cb > 0) || (mask_rb > 0 && (Pr < 1 || Pc < 1 || mask_cols < 0 || mask_cols > n)))
        return EQF_ERR_INVALID;
    // A inside B (the same rows of memory, a column offset that keeps the 32-column tiles aligned): cut once
    int aOff = -1;
    if (lda == ldb && A >= B && A - B < ldb) {
        const long long off = A - B;
        if (off % 32 == 0 && off + m <= n) aOff = (int)off;
    }
    const I8Plan p = i8Plan(m, n, k, slices, aOff);
    if (workspace_bytes < p.total) return EQF_ERR_INVALID;
    DeviceScope ds(device);
    if (!ds.ok) return EQF_ERR_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    const GemmMask mk{mask_rb, mask_cb, rblk0, Pr, pr, cblk0, Pc, pc};
    if (slices == 5) return i8Product<5>(st, C, ldc, m, n, A, lda, B, ldb, k, mk, mask_cols, aOff, ws, p);
    if (slices == 6) return i8Product<6>(st, C, ldc, m, n, A, lda, B, ldb, k, mk, mask_cols, aOff, ws, p);
    return i8Product<7>(st, C, ldc, m, n, A, lda, B, ldb, k, mk, mask_cols, aOff, ws, p);
}

int eqf_tile_downdate_i8(int device, void* stream, double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k,
    int slices, int mask_rb, void* workspace, size_t workspace_bytes) {
    if (mask_rb < 0 || (mask_rb > 0 && m != n)) return EQF_ERR_INVALID;
    return eqf_tile_gemm_tn_i8(device, stream, C, ldc, m, n, A, lda, B, ldb, k, slices, mask_rb, mask_rb, 0, 1, 0, 0, 1, 0, mask_rb > 0 ? n : 0,
        workspace, workspace_bytes);
}

int eqf_tile_propagate(int device, void* stream, double* out, const double* in, int ld, int nI, int nJ, const double* D_I,
    const double* L_I, const double* D_J, const double* L_J, const double* Sbb, const double* SbI, int ldbI, const double* SbJ,
    int ldbJ, const double* BnI, const double* BnJ, const double* R6, double T, double diag_noise, int is_diag) {
    if (!out || !in || !D_I || !L_I || !D_J || !L_J || !Sbb || !SbI || !SbJ || !BnI || !BnJ || !R6 || nI < 1 || nJ < 1 || ld < 3 * nJ ||
        ldbI < 3 * nI || ldbJ < 3 * nJ)
        return EQF_ERR_INVALID;
    DeviceScope ds(device);
    if (!ds.ok) return EQF_ERR_HIP;
    TilePropArgs a{};
    a.out = out; a.in = in; a.ld = ld; a.nI = nI; a.nJ = nJ;
    a.DI = D_I; a.LI = L_I; a.DJ = D_J; a.LJ = L_J;
    a.Sbb = Sbb; a.SbI = SbI; a.SbJ = SbJ; a.ldbI = ldbI; a.ldbJ = ldbJ;
    a.BnI = BnI; a.BnJ = BnJ;
    std::copy(R6, R6 + 6, a.R);
    a.T = T; a.diagNoise = diag_noise; a.isDiag = is_diag ? 1 : 0;
    hipLaunchKernelGGL(k_tile_propagate, dim3((nJ + 15) / 16, (nI + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    HIPC(hipGetLastError());
    return EQF_OK;
}

int eqf_tile_potrf(int device, void* stream, double* A, int ld, int n, double* drec, int* info) {
    if (!A || !drec || n < 1 || ld < n) return EQF_ERR_INVALID;
    DeviceScope ds(device);
    if (!ds.ok) return EQF_ERR_HIP;
    int rc = tileAttributes(device);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = (n + kSB - 1) / kSB;
    if (nb <= 2) {
        hipLaunchKernelGGL(k_tile_potrf, dim3(1), dim3(256), sizeof(Step64Lds), st, A, ld, n, drec, info);
    } else {
        // blocked: per 64-wide block column the diagonal block (one workgroup), the panel below it (one workgroup per 64 rows), the
        // trailing lower triangle (one workgroup per 64 x 64 tile) -- see k_tile_potrf_trail
        for (int kb = 0; kb < nb; ++kb) {
            double* Akk = A + (long long)kb * kSB * ld + kb * kSB;
            const int nk = std::min(kSB, n - kb * kSB), below = n - (kb + 1) * kSB;
            hipLaunchKernelGGL(k_tile_potrf, dim3(1), dim3(256), sizeof(Step64Lds), st, Akk, ld, nk, drec + (long long)kb * kDRec, info);
            if (below > 0) {
                hipLaunchKernelGGL(k_tile_trsm, dim3((below + kSB - 1) / kSB), dim3(256), sizeof(Step64Lds), st, Akk, ld, nk, drec + (long long)kb * kDRec,
                    Akk + (long long)kSB * ld, ld, below, 1);
                const int t = nb - kb - 1;
                hipLaunchKernelGGL(k_tile_potrf_trail, dim3(t * (t + 1) / 2), dim3(256), kTrailLdsBytes, st, A, ld, n, kb);
            }
        }
    }
    HIPC(hipGetLastError());
    return EQF_OK;
}

int eqf_tile_trsm(int device, void* stream, const double* A, int ld, int n, const double* drec, double* B, int ldb, int m, int right) {
    if (!A || !drec || !B || n < 1 || m < 1 || ld < n || ldb < (right ? n : m)) return EQF_ERR_INVALID;
    DeviceScope ds(device);
    if (!ds.ok) return EQF_ERR_HIP;
    int rc = tileAttributes(device);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tile_trsm, dim3((m + kSB - 1) / kSB), dim3(256), sizeof(Step64Lds), static_cast<hipStream_t>(stream), A, ld, n, drec, B, ldb,
        m, right ? 1 : 0);
    HIPC(hipGetLastError());
    return EQF_OK;
}

}  // extern "C"
