// C ABI of the MI355X EqF hot path (include/eqf_vio_amd.h): handle management, host-side landmark
// bookkeeping (integer id matching, eqf_vio/src/VIOFilter.cpp:211-230, :345-443) and kernel launches.
// All arithmetic of the filter runs in the HIP kernels of eqf_propagate.hpp / eqf_update.hpp /
// eqf_churn.hpp; there is no CPU fallback: without a usable GPU eqf_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/eqf_vio_amd_debug.h"  // (the public header + the test / measurement hooks this library also exports)
#include "eqf_churn.hpp"
#include "eqf_dense.hpp"
#include "eqf_device.hpp"
#include "eqf_propagate.hpp"
#include "eqf_burst.hpp"
#include "eqf_chol64.hpp"
#include "eqf_resident.hpp"
#include "eqf_update.hpp"

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
constexpr int kRing = 64;

struct ProfPair {
    int cls, key;
    hipEvent_t a, b;
};
struct ProfSample {
    int key;  // launch shape within the class (chain step index, burst length, ...): launches of one key do identical work
    float ms;
};
}  // namespace

// Ring of pinned int buffers for the small host->device uploads of the landmark bookkeeping (permutation, compaction map,
// sources of new landmarks): the host never has to drain the stream before reusing a staging buffer.
struct IntStage {
    static constexpr int kSlots = 8;
    int* host[kSlots] = {};
    hipEvent_t ev[kSlots] = {};
    int next = 0;
};

struct eqf_filter {
    int B = 0, cap = 0, device = 0, precision = 0;
    size_t esz = 8;
    eqf_settings set{};
    Params prm{};
    hipStream_t stream = nullptr;
    int nTot = 0, ld = 0;
    long long sigmaStride = 0;
    void* Sigma[2] = {nullptr, nullptr};
    Glob* g[2] = {nullptr, nullptr};
    double* p0 = nullptr;
    double* lmc = nullptr;  // [B][15][cap] per-landmark constants (C0i, chart rotation)
    double* Q[2] = {nullptr, nullptr};
    int pS = 0, pG = 0;
    // update chains
    double *SA = nullptr, *SL = nullptr, *YW = nullptr, *YO = nullptr, *EA = nullptr, *EL = nullptr, *ZW = nullptr, *ZO = nullptr;
    int ldS = 0, ldY = 0, ldE = 0, ldZ = kSB;
    long long strideDS = 0, strideDE = 0;  // diagonal-factor records of the two chains
    long long strideS = 0, strideY = 0, strideE = 0, strideZ = 0;
    double *dbgDelta = nullptr, *dbgGamma = nullptr, *dbgGammaTot = nullptr, *red = nullptr;
    int* errflag = nullptr;
    // churn scratch
    int *dMap = nullptr, *dNewN = nullptr, *dPerm = nullptr;  // (dNewN = dMap + B * cap)
    std::vector<std::vector<int>> permOnDevice;  // what dPerm holds (uploadPerm): the same landmark set in the same order needs no new copy
    double *dChord = nullptr, *dDepth2 = nullptr, *dScratch = nullptr, *dMeas = nullptr, *dOut = nullptr;
    IntStage stMap, stPerm;  // pinned rings: [B*cap + B] (map + new counts), [B*cap]
    IntStage stEdit;         // pinned ring [2*B*cap + 4*B]: keep map | permutation | counts of a k_edit launch (one upload per frame)
    int* dEdit = nullptr;    // its device image
    std::vector<int> editOnDevice;  // ... as last uploaded
    int* dEditBar = nullptr; // [B][4] k_edit's counters
    int lastBurstShape[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // eqf_debug_launch_shape: the most recent IMU burst's launch shape
    std::vector<char> restoredMark; // [B] eqf_set_state since the last reset: once every filter has been restored a hand-off time-out (bit 128) is forgiven
    // OCC2 grids from this many roles per CU on read the E-chain's tiles in Sigma itself instead of a copy made by the prep launch
    // (eqf_debug_option "e_sigma_min_percu_x10").  8.0 until late round 5 ("at 8 filters of N = 200, 6.3 roles per CU, the kernel loses what
    // the prep launch gains"); measured again behind the gated downdate: 8 filters 323.5 -> 331.3 k steps/s (prep 29.6 -> 19.2 us, kernel
    // 158.9 -> 164.1), 10 filters 345.7 -> 354.3 k; 5 / 6 / 7 filters 232.0 -> 232.9 k, 259.6 -> 262.1 k, 292.1 -> 296.8 k: every grid of the
    // two-per-CU build (chain roles per CU: 0.57 per filter of N = 200; profiles/r05_e_sigma_threshold.txt)
    double eSigmaMinPerCU = 2.4;
    // the fused burst launch up to this many workgroups per CU (eqf_debug_option "burst_fused_max_x10").  It cannot deadlock on any grid (builders
    // first, and they wait for nobody), but its bodies are the latency ones -- 4-landmark builders, one row per wavefront, 512 threads:
    // measured at N = 200 (profiles/r05_fused_oversub.txt, steps/s, two launches -> fused): 2 filters (1.2 per CU) 120.3 -> 122.3 k; 3 filters
    // (1.8) 167.7 -> 160.0 k; 4: 214.8 -> 192.9 k; 8: 331 -> 266 k.  A fused launch for batches would need the throughput bodies (16-landmark
    // builders, two / four rows per wavefront) with the flag protocol: sized at 27 us of a 262 us frame at 8 filters, not built.
    double fusedMaxPerCU = 1.25;
    int deviceEdit = 1;      // landmark-set changes and the outlier gate of a frame in ONE launch, decided on the device (eqf_debug_option "device_edit")
    double* dDepthSel = nullptr;    // [B] median scene depth selected on the device
    double *hChord = nullptr, *hMeas = nullptr, *hOut = nullptr;
    double* hChordDev = nullptr;  // device-side address of the pinned hChord
    hipEvent_t evMeas = nullptr;
    bool measPending = false;    // eqf_process_vision: the bearings sit in hMeas, their copy is visionCore's to enqueue (with k_edit's image behind them)
    // input ring for per-call records (batch > 1)
    ImuRec* dRing = nullptr;
    ImuRec* hRing = nullptr;
    hipEvent_t evRing[kRing] = {};
    bool ringUsed[kRing] = {};
    long long ringCount = 0;
    // stream mode
    ImuRec* sImu = nullptr;   // [K][B]
    ImuRec* sVis = nullptr;   // [F][B] (stamp only)
    double* sBear = nullptr;  // [F][B][nb][3]
    int sK = 0, sF = 0, sNb = 0;
    std::vector<int> sIds;
    std::vector<double> hImuStamp, hVisStamp;  // host mirrors of the stamps
    // host mirrors
    std::vector<std::vector<int>> ids;
    std::vector<double> curTime;
    std::vector<char> init;
    std::vector<char> devInit;  // the filter's lazy initialisation has been LAUNCHED (init is set when the call is queued)
    int densePropagate = 0;
    void *dF = nullptr, *dG = nullptr, *dBn = nullptr;  // dense backend: F, G = F Sigma, Bn (n x 6)
    void* dBlk = nullptr;          // split propagate path: per-landmark records [B][cap][kBlkRec] (T)
    CommonLds* dBlkCommon = nullptr;
    int streamPropagate = 1;       // split path: landmark blocks by k_riccati_stream ([no switch since round 5]: by the tile kernel)
    int splitPropagate = -1;       // -1 heuristic, 0 never, 1 always (EQF_SPLIT_PROPAGATE)
    int cholEmbed = 1;             // [no switch since round 5]: downdate + innovation lift as a launch of their own
    bool ldsAttrSet[2] = {false, false};  // hipFuncAttributeMaxDynamicSharedMemorySize applied on this handle's device
    // speculative outlier gate: the frame whose gate answer the host has not looked at yet
    struct {
        bool pending = false;
        std::vector<std::vector<int>> ids;  // measurement ids per filter
        std::vector<int> nb;
        std::vector<char> active;
        std::vector<int> nOld;              // landmarks per filter before the frame's new ones were appended
        const double* bearings = nullptr;   // device
        long long bearStride = 0;
        bool onDevice = false;              // k_edit removed the outliers itself: only the host's id lists are left to bring up to date
        std::vector<int> nKept;             // ... kept landmarks per filter (the chords in hChord are in that order)
    } gate;
    int* hGate = nullptr;        // pinned [B]: raised by k_probe
    int* hGateDev = nullptr;     // device-side address of hGate
    int* dMask = nullptr;        // [B]
    hipEvent_t evGate = nullptr;
    hipEvent_t evMask = nullptr;   // k_set_update_ok has read hGate (resolveGate's redo): recorded before the host may clear it again
    bool maskPending = false;
    int csInBurst = 1;           // bursts closed by a vision step also leave C Sigma' and S (BurstArgs::csOut; eqf_debug_option "cs_in_burst")
    bool csValid = false;        // ... and they are still what the update would compute (nothing moved a landmark since)
    int gateSpeculative = 1;     // EQF_GATE_SPECULATIVE = 0: always wait for the gate's answer before the update
    // IMU bursts (eqf_burst.hpp): processIMUData calls are queued on the host and launched together -- when the queue is
    // full, when the vision call that follows them arrives (whose integrateUpToTime joins the burst), or when the host
    // touches the handle in any other way.  burstMax = 0: every call launches at once through k_propagate.
    int burstMax = kBurstMax;      // EQF_IMU_BURST / eqf_set_imu_burst
    int burstRows = 0;             // block kernel: row landmarks per wavefront, 0 = by launch size (EQF_BURST_ROWS = 1 | 2 | 4)
    int ringAhead2 = 0;            // block kernel with two row landmarks per wave: constants requested two steps ahead -- measured 1.5-2.7 us SLOWER per burst at 4..12 filters (profiles/r06_burst_shapes.txt), kept behind eqf_debug_option "ring_ahead2"
    int burstLm = 0;               // builder: landmarks per workgroup, 0 = by launch size ([no switch since round 5])
    struct {
        int kind = 0;              // 0 nothing pending, 1 records k0 .. k0+cnt-1 of the uploaded stream, 2 inline records (one filter)
        int k0 = 0, cnt = 0;
        ImuRec inl[kBurstMax];
    } burst;
    void *dColRec = nullptr, *dRowRec = nullptr;  // per step and landmark records of k_burst_build
    BurstStep* dSteps = nullptr;
    int cholSplit = -1;            // -1 heuristic, 0 fused chain launches, 1 panel + update launches (EQF_CHOL_SPLIT)
    int cholTail = 1;              // split chain: update launches also solve the next block column ([no switch since round 5]: panel + update launches)
    int* dFlags = nullptr;         // [B][2][flagStride]: epoch flags of the in-launch hand-off of the diagonal-factor records
    int flagStride = 0;
    int updateEpoch = 0;           // one per launchUpdate
    // k_chol_resident (one launch per update while the grid fits the chip): EQF_CHOL_RESIDENT = 0 switches it off
    int cholResident = 1;
    int resOversub = -1;           // [no switch since round 5]: roles per CU up to which the resident kernel is used on a grid larger than the chip; -1 = no limit
    int resStaged = 1;             // row heads consume D[R-1] stage by stage ([no switch since round 5]: whole record after its last pivot)
    int resFoldPrep = 1;           // co-resident grid: the prep work as roles of the SAME launch (EQF_RES_FOLD_PREP = 0: k_update_prep64 launched first)
    int* dPrepFlags = nullptr;     // [B][nPrepCap]
    int nPrepCap = 0;
    int burstFused = 1;            // latency case: builder and block workgroups of a burst in ONE launch (EQF_BURST_FUSED = 0: two launches)
    int* dBuildFlags = nullptr;    // [B][nBuildCap]: steps each builder workgroup has published, + 32 * burstEpoch
    int nBuildCap = 0, burstEpoch = 0;
    bool rolesFold = false;
    int residentPerCU = -1;        // hipOccupancyMaxActiveBlocksPerMultiprocessor of k_chol_resident on this device (lazily queried)
    int eFromSigma = 1;            // split chain: block column 0 of the E-chain read straight from Sigma ([no switch since round 5]: copied by prep)
    int cholOrder = -1;            // order of the workgroup classes in an update launch, -1 = by launch size ([no switch since round 5])
    int cholStreams = 0;           // stream workgroups per filter of an update launch, 0 = by launch size ([no switch since round 5])
    int numCUs = 0;
    int nbCap = 0, wtCap = 0;
    int *dReadyA = nullptr, *dReadyY = nullptr, *dResCounters = nullptr, *dStageFlags = nullptr;
    unsigned* dTicket = nullptr;  // k_chol_resident on a grid larger than the chip: arrival tickets, [B][32] (ResArgs::ticket); ticketBase = tickets drawn per filter by earlier launches
    unsigned ticketBase = 0;
    int resTickets = 0;           // eqf_debug_option "res_tickets": 0 (default) the block index (rounds 3-5), 1 tickets on grids >= 6 x the resident slots, 2 on every grid larger than the chip (non-FOLD)
    double *dGammaPart = nullptr, *dG11Part = nullptr;
    ResRole* dRoles = nullptr;
    int resPipeHeads = -1;         // [no switch since round 5]: row heads with the pipelined panel loop (1), without (0), by grid size (-1)
    int prepOcc2 = -1;             // [no switch since round 5]: the prep launch built for two workgroups per CU (1), one (0), by launch size (-1)
    int burstOcc2 = -1;            // [no switch since round 5]: the 16-landmark builder built for two workgroups per CU (1), one (0), by launch size (-1)
    int resOcc2 = -1;              // [no switch since round 5]: k_chol_resident built for two workgroups per CU (1), one (0), by grid size (-1)
    int rolesN = -1, rolesCount = 0;  // chain shape (nbS, nbE, wtS) the role table was built for
    int rolesFront = 0;               // roles in front of the prep roles (buildRoles' `front`)
    int resFoldFront = 1;             // EQF_RES_FOLD_FRONT: dependency groups of the E-chain in front of the prep roles of a batch
    int dropRole[4] = {-1, 0, 0, 0};  // eqf_debug_drop_role: (kind, role, R, C) of the role whose workgroup leaves without publishing anything
    // profiling
    bool prof = false;
    std::vector<ProfPair> profPairs;
    std::vector<hipEvent_t> evPool;
    long long profCount[EQF_PROF_CLASSES] = {};
    double profMs[EQF_PROF_CLASSES] = {};
    std::vector<ProfSample> profSamples[EQF_PROF_CLASSES];  // per-bracket times with their launch shape
    double profOverheadMs = 0.0;  // elapsed time of an EMPTY event bracket (calibrated when profiling is switched on)
};

namespace {

template <typename F>
int profiled(eqf_filter* f, int cls, F&& launch, int key = 0) {
    if (!f->prof) {
        launch();
        return EQF_OK;
    }
    hipEvent_t a, b;
    for (hipEvent_t* e : {&a, &b}) {
        if (!f->evPool.empty()) {
            *e = f->evPool.back();
            f->evPool.pop_back();
        } else {
            HIPC(hipEventCreate(e));
        }
    }
    HIPC(hipEventRecord(a, f->stream));
    launch();
    HIPC(hipEventRecord(b, f->stream));
    f->profPairs.push_back({cls, key, a, b});
    return EQF_OK;
}

int profDrain(eqf_filter* f) {
    HIPC(hipStreamSynchronize(f->stream));
    for (auto& p : f->profPairs) {
        float ms = 0;
        HIPC(hipEventElapsedTime(&ms, p.a, p.b));
        f->profCount[p.cls]++;
        f->profMs[p.cls] += ms;
        f->profSamples[p.cls].push_back({p.key, ms});
        f->evPool.push_back(p.a);
        f->evPool.push_back(p.b);
    }
    f->profPairs.clear();
    return EQF_OK;
}

template <typename T>
int dmalloc(T** p, size_t count) {
    HIPC(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)));
    return EQF_OK;
}
template <typename T>
int hmalloc(T** p, size_t count) {
    HIPC(hipHostMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault));
    return EQF_OK;
}

int stageInit(IntStage& s, size_t count) {
    for (int i = 0; i < IntStage::kSlots; ++i) {
        int rc = hmalloc(&s.host[i], count);
        if (rc) return rc;
        HIPC(hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming));
    }
    return EQF_OK;
}
void stageFree(IntStage& s) {
    for (int i = 0; i < IntStage::kSlots; ++i) {
        if (s.host[i]) hipHostFree(s.host[i]);
        if (s.ev[i]) hipEventDestroy(s.ev[i]);
    }
}
// next free staging buffer (waits only if its previous upload, 8 uploads ago, is still in flight)
int stageAcquire(IntStage& s, int** host, int* slot) {
    *slot = s.next;
    s.next = (s.next + 1) % IntStage::kSlots;
    HIPC(hipEventSynchronize(s.ev[*slot]));
    *host = s.host[*slot];
    return EQF_OK;
}

// VIOFilter(const Settings&) initial values (VIOFilter.cpp:60-73, VIOFilter.h:45-55)
int initState(eqf_filter* f) {
    const int B = f->B;
    std::vector<Glob> g0(B);
    for (auto& g : g0) {
        std::memset(&g, 0, sizeof(Glob));
        g.P0q[0] = 1.0;
        g.Aq[0] = 1.0;
        for (int i = 0; i < 3; ++i) {
            g.bias[i] = f->set.initialOmegaBias[i];
            g.bias[3 + i] = f->set.initialAccelBias[i];
        }
        g.curTime = -1.0;
    }
    for (int p = 0; p < 2; ++p) {
        HIPC(hipMemcpyAsync(f->g[p], g0.data(), sizeof(Glob) * B, hipMemcpyHostToDevice, f->stream));
        HIPC(hipMemsetAsync(f->Sigma[p], 0, f->esz * f->sigmaStride * B, f->stream));
        HIPC(hipMemsetAsync(f->Q[p], 0, sizeof(double) * 5 * f->cap * B, f->stream));
    }
    HIPC(hipMemsetAsync(f->p0, 0, sizeof(double) * 3 * f->cap * B, f->stream));
    HIPC(hipMemsetAsync(f->errflag, 0, sizeof(int), f->stream));
    if (f->dEditBar) HIPC(hipMemsetAsync(f->dEditBar, 0, sizeof(int) * 4 * B, f->stream));  // (k_edit's counters: a launch that timed out leaves them mid-count)
    // Sigma base block diag
    std::vector<double> base(12 * 12, 0.0);
    for (int i = 0; i < 3; ++i) {
        base[i * 12 + i] = f->set.initialBiasOmegaVariance;
        base[(3 + i) * 12 + 3 + i] = f->set.initialBiasAccelVariance;
        base[(8 + i) * 12 + 8 + i] = f->set.initialVelocityVariance;
    }
    base[6 * 12 + 6] = base[7 * 12 + 7] = f->set.initialGravityVariance;
    std::vector<float> basef(base.begin(), base.end());
    for (int b = 0; b < B; ++b) {
        char* dst = static_cast<char*>(f->Sigma[0]) + f->esz * f->sigmaStride * b;
        const void* src = f->precision == EQF_PRECISION_F32 ? static_cast<const void*>(basef.data()) : static_cast<const void*>(base.data());
        HIPC(hipMemcpy2DAsync(dst, f->esz * f->ld, src, f->esz * 12, f->esz * 12, 12, hipMemcpyHostToDevice, f->stream));
    }
    HIPC(hipStreamSynchronize(f->stream));
    f->pS = f->pG = 0;
    f->gate.pending = false;
    f->measPending = false;
    f->csValid = false;
    f->editOnDevice.clear();
    f->ids.assign(B, {});
    f->curTime.assign(B, -1.0);
    f->init.assign(B, 0);
    f->devInit.assign(B, 0);
    return EQF_OK;
}

int resolveGate(eqf_filter* f);
int flushBurst(eqf_filter* f);
// Every entry point that looks at (or changes) the filter first settles what the host has deferred: the speculative gate
// of the last vision frame, then the queued IMU steps (in this order: they come after that frame).
#define GATE(f)                                                        \
    do {                                                               \
        if (hipSetDevice((f)->device) != hipSuccess) return EQF_ERR_HIP; \
        int grc_ = resolveGate(f);                                     \
        if (!grc_) grc_ = flushBurst(f);                               \
        if (grc_) return grc_;                                         \
    } while (0)

int maxN(const eqf_filter* f) {
    size_t m = 0;
    for (auto& v : f->ids) m = std::max(m, v.size());
    return int(m);
}

// Slot of the record ring holding `recs` ([B]) on the device; B == 1 passes the record inline.
int stageRecs(eqf_filter* f, const ImuRec* recs, const ImuRec** dev, int* slotOut) {
    *slotOut = -1;
    if (f->B == 1) {
        *dev = nullptr;
        return EQF_OK;
    }
    const int slot = int(f->ringCount++ % kRing);
    if (f->ringUsed[slot]) HIPC(hipEventSynchronize(f->evRing[slot]));
    std::memcpy(f->hRing + (size_t)slot * f->B, recs, sizeof(ImuRec) * f->B);
    HIPC(hipMemcpyAsync(f->dRing + (size_t)slot * f->B, f->hRing + (size_t)slot * f->B, sizeof(ImuRec) * f->B,
        hipMemcpyHostToDevice, f->stream));
    *dev = f->dRing + (size_t)slot * f->B;
    *slotOut = slot;
    return EQF_OK;
}
int releaseSlot(eqf_filter* f, int slot) {
    if (slot < 0) return EQF_OK;
    HIPC(hipEventRecord(f->evRing[slot], f->stream));
    f->ringUsed[slot] = true;
    return EQF_OK;
}

void mirrorStep(eqf_filter* f, const double* stamps, bool isImu, int* status);

// integrateUpToTime on the device + host mirror of its control flow.  devRecs == nullptr -> inline (B == 1).
int launchPropagate(eqf_filter* f, const ImuRec* devRecs, const ImuRec& inl, const double* stamps, bool isImu, bool doRiccati,
    int* status) {
    f->csValid = false;  // (a single step moves Sigma: columns of C Sigma / S left by an earlier burst are no longer the update's)
    PropArgs a{};
    a.gin = f->g[f->pG];
    a.gout = f->g[f->pG ^ 1];
    a.p0 = f->p0;
    a.Qin = f->Q[f->pG];
    a.Qout = f->Q[f->pG ^ 1];
    a.Sin = f->Sigma[f->pS];
    a.Sout = f->Sigma[f->pS ^ 1];
    a.recs = devRecs;
    a.inl = inl;
    a.errflag = f->errflag;
    a.sigmaStride = f->sigmaStride;
    a.cap = f->cap;
    a.ld = f->ld;
    a.NT = std::max(1, (maxN(f) + kTileLm - 1) / kTileLm);
    a.isImu = isImu ? 1 : 0;
    a.doRiccati = doRiccati ? 1 : 0;
    a.prm = f->prm;
    // tiles + base-block workgroup + scalar-state workgroup + NT row-tail + NT column-tail workgroups (see k_propagate)
    const dim3 grid(a.NT * a.NT + 2 + 2 * a.NT, f->B), block(256);
    int rc = EQF_OK;
    if (f->densePropagate && doRiccati) {
        // dense backend: F and Bn from the same linearisation blocks, then two MFMA GEMMs; k_propagate below only
        // does the group step and the scalar bookkeeping
        a.sigmaExternal = 1;
        const int nmax = maxN(f), nv = kLm0 + 3 * nmax, nt = (nv + 63) / 64;
        const long long bStride = (long long)f->nTot * 6;
        // (128 x 64 tiles -- twice the MFMAs per staged operand byte -- measured SLOWER than 64 x 64 at N = 1000: fp64 43.6 vs 50.4 TFLOP/s,
        // fp32 73.5 vs 89.4: fewer, fatter workgroups leave a poor last wave; removed in round 5)
        const int ntr = nt;
        rc = profiled(f, EQF_PROF_DENSE, [&] {
            auto go = [&](auto zero) {
                typedef decltype(zero) TT;
                hipLaunchKernelGGL(k_dense_build<TT>, dim3(nmax + 1, f->B), block, 0, f->stream, a, (TT*)f->dF, (TT*)f->dBn, f->sigmaStride, bStride);
                {
                    hipLaunchKernelGGL((k_dense_gemm<TT, false, 64>), dim3(nt, ntr, f->B), block, 0, f->stream, a.gin, a.recs, a.inl,
                        (const TT*)f->dF, (const TT*)a.Sin, (TT*)f->dG, (const TT*)nullptr, f->sigmaStride, bStride, f->ld, f->prm);
                    hipLaunchKernelGGL((k_dense_gemm<TT, true, 64>), dim3(nt, ntr, f->B), block, 0, f->stream, a.gin, a.recs, a.inl,
                        (const TT*)f->dG, (const TT*)f->dF, (TT*)a.Sout, (const TT*)f->dBn, f->sigmaStride, bStride, f->ld, f->prm);
                }
            };
            if (f->precision == EQF_PRECISION_F32) go(0.0f);
            else go(0.0);
        });
        if (rc) return rc;
    }
    a.blk = f->dBlk;
    a.blkCommon = f->dBlkCommon;
    // Split path (builder + lean streaming kernel) when there are enough tiles for occupancy to matter; a single small
    // filter keeps the fused single-launch kernel (one kernel boundary less per step).
    // (measured cross-over on the 256-CU part: 2500 tiles = ten per CU -- 16 filters of N = 200, N >= 800)
    const bool split = f->splitPropagate >= 0 ? f->splitPropagate != 0 : (long long)a.NT * a.NT * f->B >= 10LL * std::max(f->numCUs, 1);
    rc = profiled(f, EQF_PROF_PROPAGATE, [&] {
        if (split) {
            const dim3 bgrid((std::max(1, maxN(f)) + 63) / 64 + 1, f->B);  // landmark workgroups + the scalar-state workgroup
            // builder (blocks + G rows + group step + scalar state), then everything of Sigma by the lean streaming kernel
            // ([no switch since round 5]: by the tile kernel instead -- kept as a cross-check)
            const int nmx = std::max(1, maxN(f));
            const dim3 sgrid((nmx + 255) / 256, (nmx + kStreamRows - 1) / kStreamRows, f->B);
            if (f->precision == EQF_PRECISION_F32) {
                hipLaunchKernelGGL(k_build_blocks<float>, bgrid, dim3(128), 0, f->stream, a);
                if (f->streamPropagate) hipLaunchKernelGGL(k_riccati_stream<float>, sgrid, block, 0, f->stream, a);
                else hipLaunchKernelGGL((k_propagate<float, true>), grid, block, 0, f->stream, a);
            } else {
                hipLaunchKernelGGL(k_build_blocks<double>, bgrid, dim3(128), 0, f->stream, a);
                if (f->streamPropagate) hipLaunchKernelGGL(k_riccati_stream<double>, sgrid, block, 0, f->stream, a);
                else hipLaunchKernelGGL((k_propagate<double, true>), grid, block, 0, f->stream, a);
            }
        } else {
            // fused kernel: ... + one workgroup per 64 landmarks (group step)
            const dim3 fgrid(a.NT * a.NT + 2 + 2 * a.NT + (std::max(1, maxN(f)) + 63) / 64, f->B);
            if (f->precision == EQF_PRECISION_F32) hipLaunchKernelGGL((k_propagate<float, false>), fgrid, block, 0, f->stream, a);
            else hipLaunchKernelGGL((k_propagate<double, false>), fgrid, block, 0, f->stream, a);
        }
    });
    if (rc) return rc;
    HIPC(hipGetLastError());
    f->pG ^= 1;
    f->pS ^= 1;
    if (isImu) std::fill(f->devInit.begin(), f->devInit.end(), 1);
    mirrorStep(f, stamps, isImu, status);
    return EQF_OK;
}

// Until every filter has seen its first IMU sample (lazy initialisation, VIOFilter.cpp:122-124) calls are not queued: the
// generic schedule of the builder then only ever runs single steps, and everything after it is the fast schedule -- so
// the arithmetic of a step never depends on where the bursts are cut.
bool allDevInit(const eqf_filter* f) {
    for (char c : f->devInit)
        if (!c) return false;
    return true;
}

// K steps in two launches (eqf_burst.hpp).  Steps 0 .. K-1 read devRecs[s * recStride + b] (or inl[s], one filter); with
// visionLast the last step is the vision call's integrateUpToTime, its record visRec[b] (or inl[K-1]).
int launchBurst(eqf_filter* f, int K, const ImuRec* devRecs, long long recStride, const ImuRec* inl, bool visionLast, const ImuRec* visRec) {
    if (K <= 0) return EQF_OK;
    BurstArgs a{};
    a.gin = f->g[f->pG];
    a.gout = f->g[f->pG ^ 1];
    a.p0 = f->p0;
    a.Qin = f->Q[f->pG];
    a.Qout = f->Q[f->pG ^ 1];
    a.Sin = f->Sigma[f->pS];
    a.Sout = f->Sigma[f->pS ^ 1];
    a.recs = devRecs;
    a.recStride = recStride;
    a.visRec = visRec;
    if (inl) std::memcpy(a.inl, inl, sizeof(ImuRec) * K);
    a.K = K;
    a.visionLast = visionLast ? 1 : 0;
    a.errflag = f->errflag;
    a.sigmaStride = f->sigmaStride;
    a.cap = f->cap;
    a.ld = f->ld;
    a.colRec = f->dColRec;
    a.rowRec = f->dRowRec;
    a.steps = f->dSteps;
    a.colStep = (int)burstRecStep((long long)kColRec * f->cap);
    a.rowStep = (int)burstRecStep((long long)kBlkRec * f->cap);
    a.prm = f->prm;
    const int nmx = maxN(f);
    // builder: 4 landmarks per workgroup (its eight stages on eight wavefronts, shortest tick) while that launch fits the chip,
    // 16 per workgroup (four panel waves, full lanes) otherwise
    const int cus = std::max(f->numCUs, 1);
    // (the fused launch -- 4-landmark builders + one-row block workgroups -- is deadlock-free on any grid: up to fusedMaxPerCU workgroups per CU)
    int rb1 = 0;
    const bool fusedFits = f->burstFused && nmx > 0 && f->precision != EQF_PRECISION_F32 && f->dBuildFlags &&
                           (double)((nmx + 3) / 4 + ringTiles(nmx, 1, &rb1)) * f->B <= f->fusedMaxPerCU * cus;
    // (round 6: 8 per workgroup -- two panel waves, every serial stage still on a wave of its own -- while THOSE builders have a CU each)
    const int lm = f->burstLm ? f->burstLm
                              : (((long long)((nmx + 3) / 4) * f->B <= cus || fusedFits) ? 4 : ((long long)((nmx + 7) / 8) * f->B <= cus ? 8 : 16));
    const dim3 bgrid(std::max(1, (nmx + lm - 1) / lm), f->B);
    // (round 5: the 4-landmark builder built for two workgroups per CU so that 8 filters keep its short ticks -- 400 workgroups on 512 slots --
    // measured: 70.9 against 71.0 us per burst, nothing; not kept)
    const bool occ2 = f->burstOcc2 >= 0 ? f->burstOcc2 != 0 : (lm == 16 && (long long)bgrid.x * bgrid.y > cus);
    // rows per wavefront of the block kernel: one while the launch cannot fill the chip anyway (latency), four once the
    // column constants of a lane are worth sharing between several of its blocks.  (Two rows: 286 VGPRs, one wave per SIMD
    // like four rows but half their reuse -- measured slower than both at every size, N = 200 x 2..64 filters, N = 400..4000.)
    const long long waves1 = (long long)((nmx + 63) / 64) * nmx * f->B;
    // (cross-over measured on the 256-CU part at 2800 wave-blocks = 11 per CU: four filters of N = 200, N >= 400)
    // (in between -- the four-row tiles of the launch would not even give every CU two workgroups: 4 .. 12 filters of N = 200 -- two rows per
    // wavefront, twice the workgroups: 4 filters 62.8 -> 51.4 us per burst, 8: 75.7 -> 71.1, 12: 94.0 -> 89.6; 16: 101 -> 106)
    int rowTiles4 = 0;
    const int R = f->burstRows ? f->burstRows
                               : ((waves1 <= 11LL * cus || fusedFits) ? 1 : ((long long)ringTiles(nmx, 4, &rowTiles4) * f->B <= 3LL * cus / 2 ? 2 : 4));
    const dim3 rgrid(ringTiles(nmx, R, &a.ringBy), f->B);  // (the tiles on and below the diagonal)
    // every filter past its lazy initialisation (VIOFilter.cpp:122-124): the schedule with the precomputed step halves
    const bool fast = allDevInit(f);
    // the latency case in ONE launch: the block workgroups consume a step's records as soon as the builders have them in memory
    // (k_burst_fused).  Only while every workgroup of the launch has a CU of its own; fp64.
    const bool fused = f->burstFused && fast && lm == 4 && R == 1 && nmx > 0 && f->precision != EQF_PRECISION_F32 && f->dBuildFlags &&
                       fusedFits && (int)bgrid.x <= f->nBuildCap - 1088;
    if (fused) {
        if (f->burstEpoch >= (1 << 25)) {  // (flags are epoch * 32 + steps: start over long before the int wraps)
            HIPC(hipMemsetAsync(f->dBuildFlags, 0, sizeof(int) * f->nBuildCap * kFlagReplicas * f->B, f->stream));
            f->burstEpoch = 0;
        }
        a.buildFlags = f->dBuildFlags;
        a.epoch = ++f->burstEpoch;
        a.nBuild = (int)bgrid.x;
        a.nBuildCap = f->nBuildCap;
    }
    f->csValid = false;
    // (measured, round 5, steps/s with / without, profiles/r05_cs_in_burst.txt: 64 filters of N = 200 506.6 k / 497.1 k -- prep launch 117 -> 63 us,
    // block kernel +23 us; N = 1000 7965 / 7887; 16 filters 439.4 / 439.6 k, 8 filters 325.1 / 325.5 k: what the prep launch saves the block
    // kernel's extra stores cost; 2 filters, whose prep work hides inside the update launch, 119.3 / 120.5 k.  So: with the four-row tiles of
    // the throughput sizes; "cs_in_burst" = 2 forces it on every two-launch burst -- the bitwise tests do)
    if (visionLast && !fused && nmx > 0 && (f->csInBurst == 2 || (f->csInBurst && R == 4)) && f->YW && f->SA) {
        a.csOut = 1;
        a.YW = f->YW; a.SA = f->SA; a.lmc = f->lmc;
        a.ldY = f->ldY; a.ldS = f->ldS; a.strideY = f->strideY; a.strideS = f->strideS;
    }
    if (visionLast || f->lastBurstShape[6] == 0) {  // (the bursts that matter to a frame: the ones a vision step closes)
        const int shape[8] = {lm, R, fused ? 1 : 0, a.csOut, (int)bgrid.x, (int)rgrid.x, K, 0};
        std::copy(shape, shape + 8, f->lastBurstShape);
    }
    const int rc = profiled(f, EQF_PROF_BURST, [&] {
        if (fused) {
            hipLaunchKernelGGL((k_burst_fused<double>), dim3(bgrid.x + rgrid.x, f->B), dim3(kBuildThreads), 0, f->stream, a);
            return;
        }
        auto go = [&](auto zero) {
            typedef decltype(zero) TT;
            if (fast && lm == 4) hipLaunchKernelGGL((k_burst_build<TT, true, 4>), bgrid, dim3(kBuildThreads), 0, f->stream, a);
            else if (fast && lm == 8) hipLaunchKernelGGL((k_burst_build<TT, true, 8>), bgrid, dim3(kBuildThreads), 0, f->stream, a);
            else if (lm == 8) hipLaunchKernelGGL((k_burst_build<TT, false, 8>), bgrid, dim3(kBuildThreads), 0, f->stream, a);
            else if (fast && occ2) hipLaunchKernelGGL((k_burst_build<TT, true, 16, true>), bgrid, dim3(kBuildThreads), 0, f->stream, a);
            else if (fast) hipLaunchKernelGGL((k_burst_build<TT, true, 16>), bgrid, dim3(kBuildThreads), 0, f->stream, a);
            else if (lm == 4) hipLaunchKernelGGL((k_burst_build<TT, false, 4>), bgrid, dim3(kBuildThreads), 0, f->stream, a);
            else hipLaunchKernelGGL((k_burst_build<TT, false, 16>), bgrid, dim3(kBuildThreads), 0, f->stream, a);
            if (nmx > 0) {
                // (the four wavefronts of a workgroup share a step's column constants through an LDS ring)
                if (R == 1) hipLaunchKernelGGL((k_burst_riccati_ring<TT, 1>), rgrid, dim3(256), 0, f->stream, a);
                else if (R == 2 && f->ringAhead2) hipLaunchKernelGGL((k_burst_riccati_ring<TT, 2, true>), rgrid, dim3(256), 0, f->stream, a);
                else if (R == 2) hipLaunchKernelGGL((k_burst_riccati_ring<TT, 2, false>), rgrid, dim3(256), 0, f->stream, a);
                else hipLaunchKernelGGL((k_burst_riccati_ring<TT, 4>), rgrid, dim3(256), 0, f->stream, a);
            }
        };
        if (f->precision == EQF_PRECISION_F32) go(0.0f);
        else go(0.0);
    }, K);
    if (rc) return rc;
    HIPC(hipGetLastError());
    f->pG ^= 1;
    f->pS ^= 1;
    f->csValid = a.csOut != 0;
    if (K - (visionLast ? 1 : 0) > 0) std::fill(f->devInit.begin(), f->devInit.end(), 1);
    return EQF_OK;
}

// (Round 5, measured and dropped: launching the first call queued on an IDLE device at once -- one hipStreamQuery per burst -- so that a run that
// starts from a synchronised handle does not leave the device idle while a frame's calls are queued.  The query costs the host 3 us, a one-step
// burst 15 us of device time: the driver's 20-step shape 369 -> 393 us, steady state 65.0 -> 64.1 k steps/s.  scripts/short_run_trace.py)
bool burstEligible(const eqf_filter* f, bool isImu) {
    return f->burstMax > 0 && !f->densePropagate && (!isImu || !f->set.fastRiccati);
}

// launch the queued IMU steps (optionally closed by the integrateUpToTime of a vision call)
int flushBurstWith(eqf_filter* f, bool visionLast, const ImuRec* visDev, const ImuRec* visInl) {
    auto& q = f->burst;
    const int nImu = q.kind ? q.cnt : 0, K = nImu + (visionLast ? 1 : 0);
    if (K == 0) return EQF_OK;
    HIPC(hipSetDevice(f->device));
    ImuRec inl[kBurstMax];
    const ImuRec* dev = nullptr;
    if (q.kind == 1) dev = f->sImu + (size_t)q.k0 * f->B;
    else if (q.kind == 2) std::memcpy(inl, q.inl, sizeof(ImuRec) * nImu);
    if (visionLast && !visDev) inl[K - 1] = *visInl;
    q.kind = 0;
    q.cnt = 0;
    return launchBurst(f, K, dev, f->B, inl, visionLast, visDev);
}
int flushBurst(eqf_filter* f) { return f->burst.kind ? flushBurstWith(f, false, nullptr, nullptr) : EQF_OK; }

// host mirror (VIOFilter.cpp:120-131, :146-152, :207)
void mirrorStep(eqf_filter* f, const double* stamps, bool isImu, int* status) {
    for (int b = 0; b < f->B; ++b) {
        int st = EQF_OK;
        if (isImu) f->init[b] = 1;  // lazy initialisation happens on the first IMU sample (:122-124)
        if (f->curTime[b] < 0) st = EQF_SKIPPED_BEFORE_FIRST_IMU;
        else if (!(stamps[b] - f->curTime[b] > 0)) st = EQF_SKIPPED_NONPOSITIVE_DT;
        if (st == EQF_OK || isImu) f->curTime[b] = stamps[b];
        if (!isImu && st == EQF_OK && !f->init[b]) st = EQF_SKIPPED_NOT_INITIALISED;
        if (status) status[b] = st;
    }
}

UpdArgs makeUpdArgs(eqf_filter* f, const double* bearings, long long bearStride, const int* perm) {
    UpdArgs a{};
    a.g = f->g[f->pG];
    a.p0 = f->p0;
    a.lmc = f->lmc;
    a.Q = f->Q[f->pG];
    a.Sin = f->Sigma[f->pS];
    a.Sout = f->Sigma[f->pS ^ 1];
    a.sigmaStride = f->sigmaStride;
    a.cap = f->cap;
    a.ld = f->ld;
    a.bearings = bearings;
    a.bearStride = bearStride;
    a.perm = perm;
    a.SA = f->SA; a.SL = f->SL; a.YW = f->YW; a.YO = f->YO;
    a.ldS = f->ldS; a.ldY = f->ldY; a.strideS = f->strideS; a.strideY = f->strideY;
    a.EA = f->EA; a.EL = f->EL; a.ZW = f->ZW; a.ZO = f->ZO;
    a.ldE = f->ldE; a.ldZ = f->ldZ; a.strideE = f->strideE; a.strideZ = f->strideZ;
    a.dbgDelta = f->dbgDelta;
    a.dbgGamma = f->dbgGamma;
    a.dbgGammaTot = f->dbgGammaTot;
    a.red = f->red;
    a.errflag = f->errflag;
    a.resCounters = f->dResCounters;
    a.prm = f->prm;
    a.csInBurst = f->csValid ? 1 : 0;
    return a;
}

// Role table of k_chol_resident for chains of nbS / nbE block columns (wtS right-hand-side column tiles in the S-chain):
// block index = dependency order -- group s holds the workgroups whose last step consumes D[s]; they only wait for groups < s.
// fold: the prep work runs as roles of the same launch (ResArgs::nPrep), so the chains' first diagonal blocks are roles too (F0, in front).
// front (fold on a grid larger than the chip): the first diagonal blocks and the row heads / interior tiles of the E-chain's first `front`
// dependency groups go in FRONT of the prep roles (ResArgs::nFront): they read Sigma only, wait for nothing the prep roles write, and the
// E-chain -- the update's critical path -- starts at t = 0 instead of behind the prep workgroups of all filters.
int buildRoles(eqf_filter* f, int Nmax, bool fold, int front = 0) {
    const int nbS = roundUp(sDim(Nmax), kSB) / kSB, nbE = roundUp(eDim(Nmax), kSB) / kSB, wtS = roundUp(yCols(Nmax), kSB) / kSB;
    const int key = (fold ? 1 << 30 : 0) | (front << 27) | (nbS << 18) | (nbE << 9) | wtS;  // the table only changes when a chain crosses a 64-block boundary
    if (f->rolesN == key) return EQF_OK;
    std::vector<ResRole> r;
    if (fold) {
        r.push_back({1, 6, 0, 0});
        r.push_back({0, 6, 0, 0});
        for (int s = 0; s < std::min(front, nbE - 1); ++s) {
            r.push_back({1, 0, s + 1, 0});
            for (int R = s + 2; R < nbE; ++R) r.push_back({1, 1, R, s});
        }
    }
    f->rolesFront = fold ? int(r.size()) : 0;
    if (!(fold && front > 0)) f->rolesFront = 0;
    for (int s = 0; s < std::max(nbS, nbE); ++s)
        for (int kind = 1; kind >= 0; --kind) {  // the E-chain (the longer one) first
            const int nb = kind ? nbE : nbS, wt = kind ? 1 : wtS;
            if (s >= nb) continue;
            const bool moved = fold && kind == 1 && s < front;  // (its row head and interior tiles are in front; the right-hand-side tile waits for the prep roles)
            if (s + 1 < nb && !moved) r.push_back({kind, 0, s + 1, 0});
            for (int R = s + 2; R < nb && !moved; ++R) r.push_back({kind, 1, R, s});
            for (int t = 0; t < wt; ++t) r.push_back({kind, 2, t, s});
        }
    // fault injection (eqf_debug_drop_role): the matching role gets an index beyond every chain -- its workgroup returns at once (`R >= nb`)
    // and the flag it owes is never published: the hand-off time-out path of the kernel, on demand
    if (f->dropRole[0] >= 0)
        for (auto& q : r)
            if (q.kind == f->dropRole[0] && q.role == f->dropRole[1] && q.R == f->dropRole[2] && q.C == f->dropRole[3]) q.R = 1 << 20;
    HIPC(hipStreamSynchronize(f->stream));
    hipFree(f->dRoles);
    f->dRoles = nullptr;
    int rc = dmalloc(&f->dRoles, r.size());
    if (rc) return rc;
    HIPC(hipMemcpy(f->dRoles, r.data(), sizeof(ResRole) * r.size(), hipMemcpyHostToDevice));
    f->rolesN = key;
    f->rolesCount = int(r.size());
    return EQF_OK;
}

// (the FOLD build only exists for fp64: the fp32 mode keeps the prep launch)
template <typename T>
void launchFold(dim3 rg, hipStream_t st, const ResArgs& ra, bool pipeHeads, bool occ2) {
    if constexpr (std::is_same<T, double>::value) {
        if (pipeHeads && occ2) hipLaunchKernelGGL((k_chol_resident<double, true, true, true>), rg, dim3(256), kLdsRes2Bytes, st, ra);
        else if (pipeHeads) hipLaunchKernelGGL((k_chol_resident<double, true, false, true>), rg, dim3(256), sizeof(Step64Lds), st, ra);
        else hipLaunchKernelGGL((k_chol_resident<double, false, false, true>), rg, dim3(256), sizeof(Step64Lds), st, ra);
    }
}

template <typename T>
int launchUpdateT(eqf_filter* f, const double* bearings, long long bearStride, const int* perm, int Nmax) {
    UpdArgs a = makeUpdArgs(f, bearings, bearStride, perm);
    f->csValid = false;  // (consumed)
    const int B = f->B;
    // Factorisation kernels: k_chol_step64 / k_chol_resident (64-wide block columns, register-chained MFMA panel solves).  (The 32-wide
    // family of round 1 -- k_chol_step<INVERSE>, k_update_reduce, a separate prep launch -- was dominated at every size measured and has
    // been removed in round 3; the cross-checks of a factorisation are now the other launch shapes of the same mathematics -- resident
    // vs per-column launches bitwise, fused vs split chain to rounding -- and the fp64 oracle of the tests.)
    const int nb64S = roundUp(sDim(Nmax), kSB) / kSB, nb64E = roundUp(eDim(Nmax), kSB) / kSB;
    const int wt64 = roundUp(yCols(Nmax), kSB) / kSB;
    const int nblk64 = nb64S * nb64S + wt64 * nb64S + nb64E * nb64E + nb64E;
    a.pad = kSB;
    const int mp = roundUp(sDim(Nmax), a.pad), nep = roundUp(eDim(Nmax), a.pad);
    const int nv = kLm0 + 3 * Nmax;
    // prep
    const int nvPad = roundUp(std::min(nv, kLm0 + 3 * kPrepLmChunk), 16);
    const size_t perWave = size_t(2) * nvPad * sizeof(double);
    int wpb = int(std::min<size_t>(4, (150 * 1024) / perWave));
    if (wpb < 1) return EQF_ERR_CAPACITY;
    const int lmBlocks = (mp / 2 + wpb - 1) / wpb, eBlocks = nep / kNB;
    const size_t lds = perWave * wpb;
    // (function attributes are per device: remembered per handle, not per process)
    bool& attrSet = f->ldsAttrSet[0];
    bool& attrSet64 = f->ldsAttrSet[1];
    if (!attrSet) {
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_step64<float, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_step64<double, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_step64<float, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_step64<double, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<float>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<double>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<double, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<double, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<double, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsRes2Bytes));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<float, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsRes2Bytes));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<double, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsRes2Bytes));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<float, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<double, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds))));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<float, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsRes2Bytes));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_resident<double, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsRes2Bytes));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_step64<float, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTailBytes));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_step64<double, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTailBytes));
        attrSet = true;
    }
    // chains
    ChainArgs cS{}, cE{};
    cS.g = a.g; cS.A = f->SA; cS.D = f->SL; cS.W = f->YW; cS.WO = f->YO;
    cS.ldA = f->ldS; cS.ldW = f->ldY; cS.strideA = f->strideS; cS.strideD = f->strideDS; cS.strideW = f->strideY;
    cS.kind = 0;
    cE.g = a.g; cE.A = f->EA; cE.D = f->EL; cE.W = f->ZW; cE.WO = f->ZO;
    cE.ldA = f->ldE; cE.ldW = f->ldZ; cE.strideA = f->strideE; cE.strideD = f->strideDE; cE.strideW = f->strideZ;
    cE.kind = 1;
    f->updateEpoch = f->updateEpoch == 0x7fffffff ? 1 : f->updateEpoch + 1;
    cS.flags = f->dFlags; cS.strideF = 2 * f->flagStride; cS.epoch = f->updateEpoch;
    cE.flags = f->dFlags + f->flagStride; cE.strideF = 2 * f->flagStride; cE.epoch = f->updateEpoch;
    if (!attrSet64) {
#ifdef EQF_F64_STAMPS
        const int prepLdsMax = 158 * 1024;  // (the instrumented build keeps factor64's stamps in 1 KB of static LDS)
#else
        const int prepLdsMax = 160 * 1024;
#endif
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_prep64<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, prepLdsMax));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_prep64<double, false>), hipFuncAttributeMaxDynamicSharedMemorySize, prepLdsMax));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_prep64<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, prepLdsMax));
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_prep64<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, prepLdsMax));
        attrSet64 = true;
    }
    // ---- which shape the factorisation launches will have (decided here: the prep launch needs to know whether anybody reads EA's
    // block column 0)
    bool embed = nb64S < nb64E;
    for (int b = 0; b < B && embed; ++b) {
        const int Nb = int(f->ids[b].size());
        if (Nb > 0 && roundUp(sDim(Nb), kSB) >= roundUp(eDim(Nb), kSB)) embed = false;
    }
    if (!f->cholEmbed) embed = false;
    // (measured on the 256-CU part: 1500 workgroups = six per CU -- N = 200 from 8 filters on, N >= ~600)
    const bool splitChain = f->cholSplit >= 0 ? f->cholSplit != 0 : (long long)nblk64 * B >= 6LL * std::max(f->numCUs, 1);
    bool resident = false, residentFits = false, resPipeHeads = false, resOcc2 = false, resESigma = false;
    int rc = EQF_OK;
    bool fold = false;
    if (embed && f->cholResident && f->cholSplit <= 0 && f->dReadyA) {
        // co-residency of the whole grid by the occupancy calculation (one workgroup per CU with the 119 KB LDS image), not by
        // the CU count alone: the in-kernel downdate waits for workgroups with HIGHER block indices while holding its CU
        if (f->residentPerCU < 0) {
            int nblk = 0;
            HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void*>(&k_chol_resident<T>), 256, sizeof(Step64Lds)));
            f->residentPerCU = std::max(nblk, 0);
        }
        auto chainRoles = [](int nb, int wt) { return (nb - 1) + (nb - 1) * (nb - 2) / 2 + wt * nb; };
        const int rolesAll = chainRoles(nb64S, wt64) + chainRoles(nb64E, 1);
        residentFits = (long long)rolesAll * B <= (long long)f->residentPerCU * f->numCUs;
        // A co-resident grid (one small filter: the latency case): the prep work -- residuals, C Sigma, S, the lift rows, the chains' first
        // diagonal blocks -- runs as roles of the SAME launch (ResArgs::nPrep, role F0).  The E-chain's diagonal-factor chain, the critical
        // path of the update, needs nothing of it (Sigma_e is Sigma[6:, 6:]: its tiles are read straight from Sigma) and starts at t = 0
        // instead of behind a 12 us launch and a dispatch gap; the S-chain and the right-hand sides wait for the prep roles' flags.
        fold = f->resFoldPrep && residentFits && std::is_same<T, double>::value && f->cholResident < 2 && f->resPipeHeads <= 0 && f->resOcc2 <= 0 &&
               nb64E > 1 && f->dPrepFlags && lmBlocks + eBlocks <= f->nPrepCap && lds <= sizeof(Step64Lds);
        // Round 5: the same on a grid LARGER than the chip (the PIPEH / OCC2 builds with the prep roles in front; filter index fastest, so the
        // prep workgroups of all filters are dispatched first and wait for nobody).  Measured at N = 200 (profiles/r05_fold_batch.txt, steps/s,
        // prep launch -> prep roles): 2 filters 113.7 k -> 118.6 k; 4: 210.2 -> 210.7 k; 8: 316 -> 321 k (the update launch grows by what the
        // prep launch took: 166 -> 195 us -- the prep workgroups fill the chip first and the E-chains start behind them all the same); 16:
        // 436 -> 419 k; 64: 500 -> 465 k.  With the E-chain's first dependency group IN FRONT of the prep roles (buildRoles' `front`,
        // EQF_RES_FOLD_FRONT; profiles/r05_fold_front.txt): 4 filters 212 -> 219.6 k, 8 and 16 unchanged -- there the path prep -> S-chain ->
        // right-hand sides -> downdate is as long as the E-chain's, and the prep work is on it wherever it runs.  So: while prep roles + chain
        // roles together are at most four per CU (2 .. 4 filters of N = 200) -- round 6, behind the 8-landmark burst builder and this round's
        // other launches (profiles/r06_fold_batch.txt, best of three, prep launch -> prep roles): 5 filters 234.0 -> 246.0 k, 6: 270.6 -> 283.7 k,
        // 8: 340.4 -> 338.5 k, 10: 362.4 -> 357.0 k, 12: 397.5 -> 390.4 k -- so now up to SIX per CU (2 .. 6 filters of N = 200);
        // EQF_RES_FOLD_PREP=3 forces it on every batch (the bitwise test does), = 2 keeps it to co-resident grids.
        const bool foldBatch = (f->resFoldPrep == 1 || f->resFoldPrep == 3) && !residentFits && std::is_same<T, double>::value && f->cholResident < 2 &&
                               nb64E > 1 && f->dPrepFlags && lmBlocks + eBlocks <= f->nPrepCap && lds <= (size_t)kLdsRes2Bytes &&
                               (f->resFoldPrep == 3 || (long long)(lmBlocks + eBlocks + rolesAll) * B <= 6LL * f->numCUs);
        fold = fold || foldBatch;
        rc = buildRoles(f, Nmax, fold, foldBatch ? f->resFoldFront : 0);
        if (rc) return rc;
        // Beyond co-residency the grid is interleaved (filter index fastest: all filters advance together, group by group) and nothing
        // waits for later workgroups.  Its workgroups mostly wait for hand-offs, so the chip carries several per CU without slowing the
        // chains down; the downdate tiles are workgroups of their own at the end of the grid.  History of the switch-over (round 3, steps/s,
        // per-column launches -> this): first up to 10 roles per CU (2 .. 16 filters of N = 200: 89 -> 107 k, 132 -> 202 k, 226 -> 298 k,
        // 340 -> 373 k), then ONE filter up to ~26 roles per CU, whose per-column launches sit on the latency floor of the diagonal workgroup
        // (N = 600: 11.7 k -> 19.7 k, N = 1000: 5.2 k -> 6.3 k), and finally:
        // Since the build for two workgroups per CU (k_chol_resident's OCC2, late in round 3) the resident kernel wins at EVERY size measured
        // -- 24 / 32 / 64 / 96 filters of N = 200: 386 -> 447 k, 410 -> 479 k, 453 -> 509 k, 478 -> 507 k steps/s; one filter of N = 1500 / 2000 /
        // 3000 / 4000: 2.23 -> 2.85 k, 1053 -> 1335, 344 -> 406, 151 -> 174 -- so the default is "whenever its buffers exist"; the per-column
        // launches remain for EQF_CHOL_RESIDENT=0 / [no switch since round 5] and for filters whose chains are equally long (a handful of landmarks).
        const long long oversub = f->resOversub >= 0 ? f->resOversub : 1000000;
        resident = f->cholResident >= 2 || residentFits || (long long)f->rolesCount * B <= oversub * f->numCUs;
        // the builds of the kernel: row heads with the pipelined panel loop on grids larger than the chip (PIPEH), two workgroups per CU when
        // the grid is many times the chip (OCC2)
        resPipeHeads = f->resPipeHeads >= 0 ? f->resPipeHeads != 0 : !residentFits;
        const double perCU = double(f->rolesCount) * B / std::max(f->numCUs, 1);
        // (round 4, measured at N = 200: two per CU wins from 5 filters on -- 5 / 6 / 7 filters 217 -> 227 k, 242 -> 255 k, 270 -> 280 k steps/s -- and
        // loses below: 4 filters 210 -> 195 k; one filter of N = 400, three roles per CU, 33.5 -> 32.5 k)
        resOcc2 = resident && (f->resOcc2 >= 0 ? f->resOcc2 != 0 : (resPipeHeads && (perCU > 6.0 || (B >= 5 && perCU > 2.4))));
        resESigma = resOcc2 && resPipeHeads && perCU > f->eSigmaMinPerCU;
    }
    a.eFromSigma = (!resident && splitChain && f->cholTail && f->eFromSigma && f->precision != EQF_PRECISION_F32) ? 1 : 0;
    // (the OCC2 build reads the E-chain's tiles straight from Sigma: no copy in the prep launch; [no switch since round 5]: copied)
    if (resESigma && f->eFromSigma && f->precision != EQF_PRECISION_F32) a.eFromSigma = 2;
    fold = fold && resident;
    if (fold) a.eFromSigma = 2;
    // (round 5, measured and dropped: with the burst's operands in place, the landmark work as one LANE per landmark -- 128 per workgroup, the
    // stores from an LDS image -- instead of one wavefront: bit for bit the same and no faster.  Such a workgroup takes 21-31 us (33
    // uncoalesced loads per lane, then the stores) against 12.7 us for the 4-landmark ones, and a launch of 64 filters is three dispatch
    // rounds each as long as its longest workgroup: 63 -> 65 us; N = 1000 18 -> 82 us (one workgroup wrote all the padding rows).
    // profiles/r05_prep_stamps.txt)
    if (!fold) rc = profiled(f, EQF_PROF_UPDATE_PREP, [&] {
        // the landmark waves + E-chain operand + two more workgroups per filter that factor the first diagonal block of each chain
        // straight from Sigma (one launch: measured never slower than a separate factor launch, 4..64 filters)
        const bool occ2 = f->prepOcc2 >= 0 ? f->prepOcc2 != 0 : (B >= 4 && (long long)(lmBlocks + eBlocks + 2) * B > f->numCUs);
        if (occ2)
            hipLaunchKernelGGL((k_update_prep64<T, true>), dim3(lmBlocks + eBlocks + 2, B), dim3(256), std::max(lds, (size_t)kLdsFactorBytes), f->stream, a,
                cS, cE, lmBlocks, eBlocks, wpb, nvPad);
        else
            hipLaunchKernelGGL((k_update_prep64<T, false>), dim3(lmBlocks + eBlocks + 2, B), dim3(256), std::max(lds, (size_t)kLdsFactorBytes), f->stream, a,
                cS, cE, lmBlocks, eBlocks, wpb, nvPad);
    });
    if (rc) return rc;
    // downdate tiling: 64x64 tiles when they fill the chip, 32x32 tiles (4x the workgroups) for a single small filter
    const int nt64 = (nv + 63) / 64, nt32 = (nv + 31) / 32;
    const bool small = (long long)nt64 * (nt64 + 1) / 2 * B < 2LL * std::max(f->numCUs, 1);
    const int ddNt = small ? nt32 : nt64, ddTiles = ddNt * (ddNt + 1) / 2;
    bool tailLaunch = true;  // downdate / finish as a launch of their own after the chains
    {
        cS.nbMax = nb64S; cS.wtMax = wt64;
        cE.nbMax = nb64E; cE.wtMax = 1;
        const int steps = std::max(nb64S, nb64E);
        // The reductions ride along in the rhs workgroups; the downdate and the innovation lift ride along too when
        // every filter's S-chain is shorter than its E-chain (always, except for a handful of landmarks): `embed`, above.
        // Fused launches (each tile solves its own panel blocks) while a launch is bound by the serial diagonal chain;
        // panel + update launches (every panel block solved once, 2 workgroups per CU) once it is bound by throughput:
        // `splitChain`, above.
        auto blocks = [&](int k, int phase) {
            return chainBlocks64(cS.nbMax, cS.wtMax, k, phase) + chainBlocks64(cE.nbMax, cE.wtMax, k, phase);
        };
        // ONE launch for the whole factorisation part while its grid fits the chip (every workgroup resident: one small
        // filter, the latency case); the role table's block order keeps the ROLES deadlock-free even when it does not, and the
        // downdate -- the one wait for later workgroups -- then runs as a launch of its own: `resident` / `residentFits`, above.
        if (resident) {
            ResArgs ra{};
            ra.c0 = cS; ra.c1 = cE; ra.a = a;
            ra.roles = f->dRoles;
            ra.readyA = f->dReadyA; ra.readyY = f->dReadyY; ra.counters = f->dResCounters;
            ra.gammaPart = f->dGammaPart; ra.g11Part = f->dG11Part;
            ra.nbCap = f->nbCap; ra.wtCap = f->wtCap;
            ra.stageFlags = f->resStaged ? f->dStageFlags : nullptr;
            ra.eFromSigma = a.eFromSigma == 2 ? 1 : 0;
            if (fold) {
                ra.waitD0 = 3;
                ra.nPrep = lmBlocks + eBlocks;
                ra.nFront = f->rolesFront;
                ra.lmBlocks = lmBlocks;
                ra.prepWpb = wpb;
                ra.prepNvPad = nvPad;
                ra.prepFlags = f->dPrepFlags;
                ra.nPrepCap = f->nPrepCap;
            }
            // 64 x 64 downdate tiles: a tile costs the same 14 dependent chunk fetches whatever its size, and there are enough
            // finished workgroups to take one each
            // (a grid larger than what is co-resident must not wait for later workgroups while holding CUs: no-wait mode, see the kernel)
            ra.ddNt = (fold && !resPipeHeads) ? nt32 : nt64;  // (the FOLD build without PIPEH -- co-resident grids: 32 x 32 tiles, see the kernel)
            // (the downdate tiles are workgroups of their own behind the roles, also when the whole grid is co-resident: nothing in the kernel
            // waits for a higher block index)
            ra.nRoles = f->rolesCount;
            ra.errflag = f->errflag;
            const int ddGrid = ra.ddNt * (ra.ddNt + 1) / 2;  // downdate tiles as workgroups of their own behind the roles
            rc = profiled(f, EQF_PROF_CHOL_RESIDENT, [&] {
                // (row heads with the pipelined panel loop only on a grid larger than the chip: see the kernel's PIPEH)
                const bool pipeHeads = resPipeHeads;
                // (two workgroups per CU when the grid is many times the chip: see the kernel's OCC2)
                const bool occ2 = resOcc2;
                ra.nDdTiles = ddGrid;
                const int perFilter = ra.nPrep + f->rolesCount + ddGrid;
                ra.rolesPerRow = std::min(perFilter, 32768);
                const dim3 rg(B * ra.rolesPerRow, (perFilter + ra.rolesPerRow - 1) / ra.rolesPerRow);
                // Arrival tickets (ResArgs::ticket, the TICKET build): eqf_debug_option "res_tickets" 0 (default) = never, 2 = on every grid larger
                // than the chip that has its prep launch in front (not the FOLD build of 2 - 4 filters), 1 = where they cost least -- grids of at
                // least six times the resident slots (16+ filters of N = 200, N >= ~700), whose workgroups are dispatched long before they are needed;
                // on lightly oversubscribed grids a role is dispatched just in time and the ticket's round trip (~2 us) lands on the critical
                // path at every dependency hop: +12.8 / +7.4 / +14.6 / +3.2 / +8.6 us per update at 2 / 4 / 6 / 8 / 12 filters, +1.1 % at 64, +0.7 %
                // at N = 1000 in one binary (profiles/r06_tickets_ab.txt), 2.3 % against round 5's library on the same box.  OFF by default: the
                // block index with the time-outs as its guard, as in rounds 3 - 5 -- the guarantee is there for whoever wants to pay for it.
                const long long slots = (long long)std::max(f->numCUs, 1) * (occ2 ? 2 : 1);
                const bool tickets = f->resTickets == 2 || (f->resTickets == 1 && (long long)rg.x * rg.y >= 6 * slots);
                const bool ticketBuild = pipeHeads && !fold && tickets && f->dTicket;
                if (ticketBuild) {  // (every workgroup of the launch draws exactly one ticket, padding workgroups included)
                    ra.ticket = f->dTicket;
                    ra.ticketBase = f->ticketBase;
                    f->ticketBase += ra.rolesPerRow * rg.y;  // (per filter)
                }
                if (ticketBuild && occ2) hipLaunchKernelGGL((k_chol_resident<T, true, true, false, true>), rg, dim3(256), kLdsRes2Bytes, f->stream, ra);
                else if (ticketBuild) hipLaunchKernelGGL((k_chol_resident<T, true, false, false, true>), rg, dim3(256), sizeof(Step64Lds), f->stream, ra);
                else if (fold && pipeHeads) launchFold<T>(rg, f->stream, ra, true, occ2);
                else if (pipeHeads && occ2) hipLaunchKernelGGL((k_chol_resident<T, true, true>), rg, dim3(256), kLdsRes2Bytes, f->stream, ra);
                else if (pipeHeads) hipLaunchKernelGGL((k_chol_resident<T, true>), rg, dim3(256), sizeof(Step64Lds), f->stream, ra);
                else if (fold) launchFold<T>(rg, f->stream, ra, false, false);
                else hipLaunchKernelGGL((k_chol_resident<T, false>), rg, dim3(256), sizeof(Step64Lds), f->stream, ra);
            });
            if (rc) return rc;
        } else if (splitChain && f->cholTail) {
            // one launch per block column: the panel launch of column 0, then update launches that also solve column k+1
            // (k_chol_step64<T, 3>); the S-chain's right-hand sides are complete after launch nb64S - 2, the downdate joins
            // launch nb64S - 1 (or runs on its own below when there is none)
            rc = profiled(f, EQF_PROF_CHOL_STEP, [&] {
                hipLaunchKernelGGL((k_chol_step64<T, 1>), dim3(blocks(0, 1), B), dim3(256), sizeof(Step64Lds), f->stream, cS, cE, a, 0, 0, 0,
                    embed ? 1 : 0, f->errflag);
            }, 1000);
            if (rc) return rc;
            // (A downdate with 128 x 128 tiles -- half the Y traffic per flop -- as a launch of its own was tried here and measured
            // SLOWER: 788 vs 516 us for 64 filters, 48 vs 29.5 ms at N = 4000; 352 registers leave one wave per SIMD.  The 64-wide
            // tiles are not bandwidth-bound: they run at half the fp64 MFMA peak in executed flops.)
            bool ddDone = false;
            for (int k = 0; k + 1 < steps; ++k) {
                const int dd = (embed && k == nb64S - 1) ? ddTiles : 0;
                if (dd) ddDone = true;
                // workgroups per filter: 2 diagonal + the tails of block column k+1 + nStream streams over the pure updates
                // (about two per CU over the whole launch; four for ONE large filter: N = 4000 150 -> 154 steps/s) + the downdate tiles; see step3Counts
                int tS, uS, tE, uE;
                step3Counts(cS.nbMax, cS.wtMax, k, &tS, &uS);
                step3Counts(cE.nbMax, cE.wtMax, k, &tE, &uE);
                const int nStream = f->cholStreams > 0 ? std::min(f->cholStreams, std::max(uS + uE, 1))
                                                       : std::min(uS + uE, std::max(1, ((B == 1 ? 4 : 2) * f->numCUs + B - 1) / B));
                const int tailsLast = f->cholOrder >= 0 ? f->cholOrder : 1;
                rc = profiled(f, dd ? EQF_PROF_CHOL_DD : EQF_PROF_CHOL_STEP, [&] {
                    hipLaunchKernelGGL((k_chol_step64<T, 3>), dim3(2 + tS + tE + nStream + dd, B), dim3(256), kLdsTailBytes, f->stream, cS, cE, a,
                        k, dd ? ddNt : 0, small ? 1 : 0, embed ? 1 : 0, f->errflag, nStream, tailsLast);
                }, k);
                if (rc) return rc;
            }
            if (embed && !ddDone) embed = false;  // (cannot happen while nb64S < nb64E: kept for safety -> tail launch below)
        } else
        for (int k = 0; k < steps; ++k) {
            {
                const int dd = (embed && k == nb64S) ? ddTiles : 0;
                rc = profiled(f, dd ? EQF_PROF_CHOL_DD : EQF_PROF_CHOL_STEP, [&] {
                    hipLaunchKernelGGL((k_chol_step64<T, 0>), dim3(blocks(k, 0) + dd, B), dim3(256), sizeof(Step64Lds), f->stream, cS, cE, a, k,
                        dd ? ddNt : 0, small ? 1 : 0, embed ? 1 : 0, f->errflag);
                }, k);
            }
            if (rc) return rc;
        }
        tailLaunch = !embed;
    }
    if (tailLaunch) {
        // the last workgroup of the launch runs the (independent) innovation-lift / group-update part
        rc = profiled(f, EQF_PROF_DOWNDATE, [&] {
            if (small) hipLaunchKernelGGL((k_downdate<T, 32>), dim3(ddTiles + 1, B), dim3(256), (downdateLdsBytes<T, 32>()), f->stream, a, nt32, 1);
            else hipLaunchKernelGGL((k_downdate<T, 64>), dim3(ddTiles + 1, B), dim3(256), (downdateLdsBytes<T, 64>()), f->stream, a, nt64, 1);
        });
        if (rc) return rc;
    }
    HIPC(hipGetLastError());
    f->pS ^= 1;
    return EQF_OK;
}

int launchUpdate(eqf_filter* f, const double* bearings, long long bearStride, const int* perm, int Nmax) {
    return f->precision == EQF_PRECISION_F32 ? launchUpdateT<float>(f, bearings, bearStride, perm, Nmax)
                                             : launchUpdateT<double>(f, bearings, bearStride, perm, Nmax);
}

// Batch-wide compaction with host keep-lists keep[b] = old indices that survive (ascending).
int compact(eqf_filter* f, const std::vector<std::vector<int>>& keep) {
    const int B = f->B, cap = f->cap;
    f->csValid = false;
    int* h = nullptr;
    int slot = 0;
    int rcs = stageAcquire(f->stMap, &h, &slot);
    if (rcs) return rcs;
    int nmax = 0;
    for (int b = 0; b < B; ++b) {
        h[(size_t)B * cap + b] = int(keep[b].size());
        nmax = std::max(nmax, int(keep[b].size()));
        std::copy(keep[b].begin(), keep[b].end(), h + (size_t)b * cap);
    }
    HIPC(hipMemcpyAsync(f->dMap, h, sizeof(int) * ((size_t)B * cap + B), hipMemcpyHostToDevice, f->stream));  // (map + new counts: one copy)
    HIPC(hipEventRecord(f->stMap.ev[slot], f->stream));
    const int nvn = kLm0 + 3 * nmax;
    int rc = profiled(f, EQF_PROF_CHURN, [&] {
        const dim3 grid((nvn + 255) / 256, nvn, B);
        if (f->precision == EQF_PRECISION_F32)
            hipLaunchKernelGGL(k_compact<float>, grid, dim3(256), 0, f->stream, f->g[f->pG], f->dMap, f->dNewN, cap,
                static_cast<const float*>(f->Sigma[f->pS]), static_cast<float*>(f->Sigma[f->pS ^ 1]), f->sigmaStride, f->ld, f->p0, f->Q[f->pG], f->lmc,
                f->dScratch);
        else
            hipLaunchKernelGGL(k_compact<double>, grid, dim3(256), 0, f->stream, f->g[f->pG], f->dMap, f->dNewN, cap,
                static_cast<const double*>(f->Sigma[f->pS]), static_cast<double*>(f->Sigma[f->pS ^ 1]), f->sigmaStride, f->ld, f->p0, f->Q[f->pG], f->lmc,
                f->dScratch);
    });
    if (rc) return rc;
    HIPC(hipGetLastError());
    f->pS ^= 1;
    return EQF_OK;
}

// perm[b][i] = index into the measurement of state landmark i (or -1)
int uploadPerm(eqf_filter* f, const std::vector<std::vector<int>>& perm) {
    const int B = f->B, cap = f->cap;
    // A landmark that was removed as an outlier and comes back is appended at the END of the state: from then on the state's order differs
    // from the measurement's on every frame, with the SAME permutation as long as the set does not change.  Each copy is a blit launch on
    // the stream (4 us + its boundaries; the gate and the update both ask for the permutation: 2.4 copies per frame in the churn leg).
    if (perm == f->permOnDevice) return EQF_OK;
    int* h = nullptr;
    int slot = 0;
    int rc = stageAcquire(f->stPerm, &h, &slot);
    if (rc) return rc;
    for (int b = 0; b < B; ++b) {
        std::fill(h + (size_t)b * cap, h + (size_t)(b + 1) * cap, -1);
        std::copy(perm[b].begin(), perm[b].end(), h + (size_t)b * cap);
    }
    f->permOnDevice.clear();  // (not valid if the copy cannot be enqueued)
    HIPC(hipMemcpyAsync(f->dPerm, h, sizeof(int) * B * cap, hipMemcpyHostToDevice, f->stream));
    HIPC(hipEventRecord(f->stPerm.ev[slot], f->stream));
    f->permOnDevice = perm;
    return EQF_OK;
}

// chord errors (when bearings are given) and squared depths of every landmark of the current estimate -> dChord, dDepth2;
// readback = true also copies them to the host and waits (only the outlier gate needs that: the host owns the ids)
int probe(eqf_filter* f, const double* bearings, long long bearStride, bool withPerm, bool readback, bool speculative = false) {
    const int B = f->B, cap = f->cap;
    const int nmax = std::max(1, maxN(f));
    // with readback the kernel writes the chords straight into pinned host memory (no copy command behind it)
    double* chordDst = (readback || speculative) ? f->hChordDev : f->dChord;  // (speculative: resolveGate reads them if the gate tripped)
    int rc = profiled(f, EQF_PROF_CHURN, [&] {
        hipLaunchKernelGGL(k_probe, dim3((nmax + 127) / 128, B), dim3(128), 0, f->stream, f->g[f->pG], f->p0, f->Q[f->pG], cap, bearings,
            bearStride, withPerm ? f->dPerm : nullptr, chordDst, f->dDepth2, f->set.outlierThreshold, speculative ? f->hGateDev : nullptr);
    });
    if (rc) return rc;
    if (readback) HIPC(hipStreamSynchronize(f->stream));
    return EQF_OK;
}

// processVisionData after integrateUpToTime: bookkeeping + update.  measIds[b] ascending ids of filter b,
// device bearings at bearings + b*bearStride.  active[b] = integration succeeded && initialised.
int visionCore(eqf_filter* f, const std::vector<const int*>& measIds, const std::vector<int>& nb, const double* bearings,
    long long bearStride, const std::vector<char>& active, int* status, const std::vector<std::vector<char>>* gated = nullptr) {
    // gated (resolveGate's redo of a frame whose speculative gate tripped): the outliers are known and already removed, (*gated)[b][k] marks
    // their measurement entries -- the gate is not evaluated again
    const int B = f->B, cap = f->cap;
    // (per-call API: the bearings are still in pinned host memory -- whoever enqueues the first consumer enqueues their copy)
    auto flushMeas = [&]() -> int {
        if (!f->measPending) return EQF_OK;
        f->measPending = false;
        HIPC(hipMemcpyAsync(f->dMeas, f->hMeas, sizeof(double) * 3 * cap * B, hipMemcpyHostToDevice, f->stream));
        HIPC(hipEventRecord(f->evMeas, f->stream));
        return EQF_OK;
    };
    // ---- removeOldLandmarks (VIOFilter.cpp:393-419): state ids absent from the measurement
    bool anyLost = false;
    std::vector<std::vector<int>> keep(B);
    for (int b = 0; b < B; ++b) {
        auto& sid = f->ids[b];
        keep[b].resize(sid.size());
        for (size_t i = 0; i < sid.size(); ++i) keep[b][i] = int(i);
        if (!active[b]) continue;
        std::vector<int> kept;
        for (size_t i = 0; i < sid.size(); ++i)
            if (std::binary_search(measIds[b], measIds[b] + nb[b], sid[i])) kept.push_back(int(i));
        if (kept.size() != sid.size()) {
            anyLost = true;
            keep[b] = kept;
        }
    }
    // ---- round 5: the whole landmark bookkeeping of the frame in one launch (k_edit, eqf_churn.hpp) -- the host's keep list, the outlier
    // gate evaluated AND acted upon on the device, the new landmarks -- with one upload, and no frame is ever redone.  Not for a redo
    // itself, filters beyond kEditMax landmarks, a gate whose previous answer is still pending or switched to the synchronous mode, and
    // -- gate armed -- filters so small that losing a landmark could make their two chains equally long (the update's launch shape is
    // chosen from the host's count).
    {
        const bool gateArmed = !gated && f->set.outlierThreshold < 2.0 && maxN(f) > 0;
        bool ok = f->deviceEdit && !gated && f->dEdit && (!gateArmed || (f->gateSpeculative && !f->gate.pending));
        bool anyFresh = false;
        for (int b = 0; b < B && ok; ++b) {
            if (int(f->ids[b].size()) > kEditMax) ok = false;
            if (!active[b]) continue;
            if (gateArmed && nb[b] < kEditSafeN) ok = false;  // (kept + new landmarks = the measurement's entries)
            if (nb[b] > int(keep[b].size())) anyFresh = true;  // (every kept id is in the measurement)
        }
        if (ok && (anyLost || anyFresh || gateArmed)) {
            // (per-call API: the image rides behind the bearings, one copy for both; stream API: a staging ring of its own)
            const bool withMeas = f->measPending;
            int* h = nullptr;
            int slot = 0;
            int rc = EQF_OK;
            if (withMeas) h = reinterpret_cast<int*>(f->hMeas + (size_t)3 * cap * B);
            else rc = stageAcquire(f->stEdit, &h, &slot);
            if (rc) return rc;
            std::vector<int> nKept(B, 0);
            // (the new id lists and the "no bearings left" verdicts are committed to the handle only once k_edit is in the stream: an error on
            // the way there -- capacity, a failed copy, a failed launch -- leaves the host's lists describing what the device still holds)
            std::vector<std::vector<int>> newIds(B);
            std::vector<int> skipped;
            bool anyWork = false;
            int Nmax = 0;
            for (int b = 0; b < B; ++b) {
                int* hm = h + (size_t)b * cap;
                int* hp = h + (size_t)(B + b) * cap;
                int* hc = h + (size_t)2 * B * cap + 4 * b;
                const int nK = int(keep[b].size());
                nKept[b] = nK;
                std::copy(keep[b].begin(), keep[b].end(), hm);
                std::fill(hm + nK, hm + cap, -1);
                std::vector<int> nid;
                for (int o : keep[b]) nid.push_back(f->ids[b][o]);
                int nNew = 0;
                std::fill(hp, hp + cap, -1);
                if (active[b]) {
                    std::vector<char> used(nb[b], 0);
                    for (int j = 0; j < nK; ++j) {
                        const int k = int(std::lower_bound(measIds[b], measIds[b] + nb[b], nid[j]) - measIds[b]);
                        hp[j] = k;
                        used[k] = 1;
                    }
                    if (nK + (nb[b] - nK) > cap) return EQF_ERR_CAPACITY;  // (cannot happen: every entry point checks nb <= capacity)
                    for (int k = 0; k < nb[b]; ++k)
                        if (!used[k]) {
                            hp[nK + nNew++] = k;
                            nid.push_back(measIds[b][k]);
                        }
                }
                hc[0] = nK; hc[1] = nNew; hc[2] = (gateArmed && active[b]) ? 1 : 0; hc[3] = 0;
                const bool empty = nid.empty();
                const int nNow = int(nid.size());
                newIds[b] = std::move(nid);
                if (!active[b]) continue;
                if (empty) {
                    skipped.push_back(b);
                    continue;
                }
                anyWork = true;
                Nmax = std::max(Nmax, nNow);
            }
            // (a fixed set behind an armed gate: the same image every frame, nothing to upload)
            const size_t nInts = (size_t)2 * B * cap + 4 * B;
            const int* devImage = f->dEdit;
            if (withMeas) {
                f->measPending = false;
                HIPC(hipMemcpyAsync(f->dMeas, f->hMeas, sizeof(double) * 3 * cap * B + sizeof(int) * nInts, hipMemcpyHostToDevice, f->stream));
                HIPC(hipEventRecord(f->evMeas, f->stream));
                devImage = reinterpret_cast<const int*>(f->dMeas + (size_t)3 * cap * B);
            } else if (f->editOnDevice.size() != nInts || !std::equal(h, h + nInts, f->editOnDevice.begin())) {
                f->editOnDevice.clear();
                HIPC(hipMemcpyAsync(f->dEdit, h, sizeof(int) * nInts, hipMemcpyHostToDevice, f->stream));
                HIPC(hipEventRecord(f->stEdit.ev[slot], f->stream));
                f->editOnDevice.assign(h, h + nInts);
            }
            if (gateArmed) {
                if (f->maskPending) {
                    HIPC(hipEventSynchronize(f->evMask));
                    f->maskPending = false;
                }
                std::fill(f->hGate, f->hGate + B, 0);
            }
            EditArgs ea{};
            ea.g = f->g[f->pG];
            ea.in = devImage;
            ea.permOut = f->dPerm;
            ea.B = B; ea.cap = cap;
            ea.bearings = bearings; ea.bearStride = bearStride;
            ea.gateThr = f->set.outlierThreshold;
            ea.gateFlag = gateArmed ? f->hGateDev : nullptr;
            ea.chordOut = gateArmed ? f->hChordDev : nullptr;
            ea.depthDefault = f->set.initialSceneDepth; ea.pointVar = f->set.initialPointVariance;
            ea.p0 = f->p0; ea.Q = f->Q[f->pG]; ea.lmc = f->lmc;
            ea.errflag = f->errflag;
            ea.Scur = f->Sigma[f->pS]; ea.Soth = f->Sigma[f->pS ^ 1];
            ea.sigmaStride = f->sigmaStride; ea.ld = f->ld;
            ea.hostFlip = anyLost ? 1 : 0;
            ea.bar = f->dEditBar;
            // (co-resident when an outliers-only compaction may have to wait for its workgroups; a frame whose Sigma the host knows to move
            // anyway waits for nobody: a workgroup per four rows)
            const int coRes = std::max(1, std::min(64, std::max(f->numCUs, 1) / B));
            const int G = (anyLost || !gateArmed) ? std::max(coRes, std::min(1024, (kLm0 + 3 * Nmax + 3) / 4)) : coRes;  // (no gate: nothing can trip)
            rc = profiled(f, EQF_PROF_CHURN, [&] {
                if (f->precision == EQF_PRECISION_F32) hipLaunchKernelGGL(k_edit<float>, dim3(G, B), dim3(256), 0, f->stream, ea);
                else hipLaunchKernelGGL(k_edit<double>, dim3(G, B), dim3(256), 0, f->stream, ea);
            });
            if (rc) return rc;
            HIPC(hipGetLastError());
            for (int b = 0; b < B; ++b) f->ids[b] = std::move(newIds[b]);
            if (status)
                for (int b : skipped) status[b] = EQF_SKIPPED_NO_BEARINGS;
            if (anyLost) f->pS ^= 1;
            f->csValid = false;
            f->permOnDevice.clear();  // (dPerm now holds what k_edit made of the upload)
            if (gateArmed) {
                HIPC(hipEventRecord(f->evGate, f->stream));
                f->gate.pending = true;
                f->gate.onDevice = true;
                f->gate.nKept = nKept;
                f->gate.active = active;
                f->gate.bearings = bearings;
                f->gate.bearStride = bearStride;
            }
            if (!anyWork) return EQF_OK;
            return launchUpdate(f, bearings, bearStride, f->dPerm, Nmax);
        }
    }
    {
        int rc = flushMeas();
        if (rc) return rc;
    }
    if (anyLost) {
        int rc = compact(f, keep);
        if (rc) return rc;
        for (int b = 0; b < B; ++b) {
            std::vector<int> nid;
            for (int o : keep[b]) nid.push_back(f->ids[b][o]);
            f->ids[b] = nid;
        }
    }
    // ---- matchMeasurementsToState (:211-230): perm[b][i] = measurement index of state landmark i
    std::vector<std::vector<int>> perm(B);
    auto buildPerm = [&]() {
        for (int b = 0; b < B; ++b) {
            perm[b].assign(f->ids[b].size(), -1);
            if (!active[b]) continue;
            for (size_t i = 0; i < f->ids[b].size(); ++i) {
                const int* it = std::lower_bound(measIds[b], measIds[b] + nb[b], f->ids[b][i]);
                perm[b][i] = int(it - measIds[b]);
            }
        }
    };
    buildPerm();
    // ---- removeOutliers (:429-443).  A chord between unit vectors never exceeds 2.
    std::vector<std::vector<char>> dropped(B);  // measurement indices erased together with their landmark
    for (int b = 0; b < B; ++b) dropped[b].assign(nb[b], 0);
    if (gated) dropped = *gated;
    const bool gateOn = !gated && f->set.outlierThreshold < 2.0 && maxN(f) > 0;
    // Speculative gate: the probe decides on the device, the frame's new landmarks and the update are enqueued without waiting for the
    // answer, and a frame that did have an outlier is redone the slow way by resolveGate() the next time the host touches the handle
    // (which first takes the frame's new landmarks out again: the reference removes the outliers BEFORE it initialises new landmarks
    // at the median depth, VIOFilter.cpp:429-443 then :345-391).
    bool speculate = gateOn && f->gateSpeculative && !f->gate.pending;  // (lost landmarks are already compacted away)
    bool depthFresh = false;  // dDepth2 holds the squared depths of the CURRENT landmark set
    if (speculate) {
        // the permutation the update will use -- the frame's new landmarks appended in measurement order -- is uploaded once, now: the
        // probe reads its first N entries
        std::vector<std::vector<int>> permFull = perm;
        for (int b = 0; b < B; ++b) {
            if (!active[b]) continue;
            std::vector<char> used(nb[b], 0);
            for (int k : perm[b]) used[k] = 1;
            for (int k = 0; k < nb[b]; ++k)
                if (!used[k]) permFull[b].push_back(k);
        }
        bool identityPerm = true;
        for (int b = 0; b < B && identityPerm; ++b)
            for (size_t i = 0; i < permFull[b].size(); ++i)
                if (permFull[b][i] != int(i)) {
                    identityPerm = false;
                    break;
                }
        int rc = identityPerm ? EQF_OK : uploadPerm(f, permFull);
        if (rc) return rc;
        if (f->maskPending) {  // (a redo's k_set_update_ok reads hGate)
            HIPC(hipEventSynchronize(f->evMask));
            f->maskPending = false;
        }
        std::fill(f->hGate, f->hGate + B, 0);
        rc = probe(f, bearings, bearStride, !identityPerm, false, true);
        if (rc) return rc;
        depthFresh = true;
        HIPC(hipEventRecord(f->evGate, f->stream));
        f->gate.pending = true;
        f->gate.onDevice = false;
        f->gate.ids.assign(B, {});
        f->gate.nOld.assign(B, 0);
        for (int b = 0; b < B; ++b) {
            f->gate.ids[b].assign(measIds[b], measIds[b] + nb[b]);
            f->gate.nOld[b] = int(f->ids[b].size());
        }
        f->gate.nb = nb;
        f->gate.active = active;
        f->gate.bearings = bearings;
        f->gate.bearStride = bearStride;
    } else if (gateOn) {
        bool identityPerm = true;
        for (int b = 0; b < B && identityPerm; ++b)
            for (size_t i = 0; i < perm[b].size(); ++i)
                if (perm[b][i] != int(i)) {
                    identityPerm = false;
                    break;
                }
        int rc = identityPerm ? EQF_OK : uploadPerm(f, perm);
        if (rc) return rc;
        rc = probe(f, bearings, bearStride, !identityPerm, true);
        if (rc) return rc;
        bool anyOut = false;
        for (int b = 0; b < B; ++b) {
            keep[b].clear();
            for (size_t i = 0; i < f->ids[b].size(); ++i) {
                const bool out = active[b] && f->hChord[(size_t)b * cap + i] > f->set.outlierThreshold;
                if (out) {
                    anyOut = true;
                    dropped[b][perm[b][i]] = 1;
                } else {
                    keep[b].push_back(int(i));
                }
            }
        }
        if (anyOut) {
            rc = compact(f, keep);
            if (rc) return rc;
            for (int b = 0; b < B; ++b) {
                std::vector<int> nid;
                for (int o : keep[b]) nid.push_back(f->ids[b][o]);
                f->ids[b] = nid;
            }
            buildPerm();
        }
        depthFresh = !anyOut;  // (the probe also left the squared depths -- of the set before any removal)
    }
    // ---- addNewLandmarks (:345-391)
    bool needDepth = false;
    std::vector<std::vector<int>> fresh(B);  // measurement indices of new landmarks, in measurement order
    for (int b = 0; b < B; ++b) {
        if (!active[b]) continue;
        {
            // (every state id is in the measurement by now: perm marks the measurement entries that have a landmark)
            std::vector<char> used(nb[b], 0);
            for (int k : perm[b])
                if (k >= 0 && k < nb[b]) used[k] = 1;
            for (int k = 0; k < nb[b]; ++k)
                if (!used[k] && !dropped[b][k]) fresh[b].push_back(k);
        }
        if (!fresh[b].empty()) {
            // (cannot happen: every entry point checks nb <= capacity and strictly ascending ids before any effect)
            if (f->ids[b].size() + fresh[b].size() > (size_t)cap) return EQF_ERR_CAPACITY;
            if (!f->ids[b].empty()) needDepth = true;
        }
    }
    // the host's ids first, then the permutation the update will use (new landmarks at the end, in measurement order): uploaded once, and
    // k_append finds the bearing of the j-th new landmark in it
    std::vector<int> nOldV(B, 0);
    for (int b = 0; b < B; ++b) {
        nOldV[b] = int(f->ids[b].size());
        for (int k : fresh[b]) f->ids[b].push_back(measIds[b][k]);
    }
    buildPerm();
    bool identity = true, anyWork = false;
    int Nmax = 0;
    for (int b = 0; b < B; ++b) {
        if (!active[b]) continue;
        if (f->ids[b].empty()) {
            if (status) status[b] = EQF_SKIPPED_NO_BEARINGS;
            continue;
        }
        anyWork = true;
        Nmax = std::max(Nmax, int(f->ids[b].size()));
        for (size_t i = 0; i < perm[b].size(); ++i)
            if (perm[b][i] != int(i)) identity = false;
    }
    if (anyWork && !identity) {
        int rc = uploadPerm(f, perm);
        if (rc) return rc;
    }
    bool medianLaunch = false;
    if (needDepth) {
        // squared depths of the current estimate on the device (adding landmarks needs no readback); their median is selected by k_append
        // itself, or by a launch of its own for sets too large for that
        // (k_append computes the squared depths itself, round 5: no probe launch in front of it; only sets too large for its LDS take the
        // probe + the selection launch)
        int rc = EQF_OK;
        for (int b = 0; b < B; ++b)
            if (!fresh[b].empty() && nOldV[b] > kMedianInAppend) medianLaunch = true;
        if (medianLaunch && !depthFresh) rc = probe(f, nullptr, 0, false, false);
        if (rc) return rc;
        if (medianLaunch) {
            int nmx = 1;
            for (int b = 0; b < B; ++b) nmx = std::max(nmx, nOldV[b]);
            // (k_median_depth reads N from the device state, which still holds the old counts)
            rc = profiled(f, EQF_PROF_CHURN, [&] {
                hipLaunchKernelGGL(k_median_depth, dim3((nmx + 255) / 256, B), dim3(256), 0, f->stream, f->g[f->pG], f->dDepth2, cap, f->dDepthSel);
            });
            if (rc) return rc;
        }
    }
    for (int b = 0; b < B; ++b) {
        if (fresh[b].empty()) continue;
        f->csValid = false;
        const int nOld = nOldV[b], nNew = int(fresh[b].size());
        const long long work = (long long)3 * nNew * (kLm0 + 3 * (nOld + nNew)) * 2;
        const int blocks = int(std::min<long long>(1024, (work + 255) / 256));
        const double* depthSel = medianLaunch ? f->dDepthSel : nullptr;
        const int* permDev = identity ? nullptr : f->dPerm;
        int rc = profiled(f, EQF_PROF_CHURN, [&] {
            if (f->precision == EQF_PRECISION_F32)
                hipLaunchKernelGGL(k_append<float>, dim3(std::max(1, blocks)), dim3(256), 0, f->stream, f->g[f->pG], b, nOld, nNew, depthSel, (const double*)nullptr,
                    f->set.initialSceneDepth, f->set.initialPointVariance, cap, bearings + (long long)b * bearStride, permDev, f->p0,
                    f->Q[f->pG], f->lmc, f->errflag, static_cast<float*>(f->Sigma[f->pS]), f->sigmaStride, f->ld);
            else
                hipLaunchKernelGGL(k_append<double>, dim3(std::max(1, blocks)), dim3(256), 0, f->stream, f->g[f->pG], b, nOld, nNew, depthSel, (const double*)nullptr,
                    f->set.initialSceneDepth, f->set.initialPointVariance, cap, bearings + (long long)b * bearStride, permDev, f->p0,
                    f->Q[f->pG], f->lmc, f->errflag, static_cast<double*>(f->Sigma[f->pS]), f->sigmaStride, f->ld);
        });
        if (rc) return rc;
        HIPC(hipGetLastError());
    }
    // ---- the update proper (:258-297)
    if (!anyWork) return EQF_OK;
    return launchUpdate(f, bearings, bearStride, identity ? nullptr : f->dPerm, Nmax);
}

// Look at the answer of a speculative outlier gate.  No outlier (the common case): nothing to do.  Otherwise the filters
// that saw one had their update switched off on the device; the frame is redone for them the synchronous way (remove the
// outliers, then update), exactly what a non-speculative call would have done at the time.
int resolveGate(eqf_filter* f) {
    if (!f->gate.pending) return EQF_OK;
    HIPC(hipSetDevice(f->device));
    HIPC(hipEventSynchronize(f->evGate));
    f->gate.pending = false;
    const int B = f->B, cap = f->cap;
    if (f->gate.onDevice) {
        // k_edit took the outliers out before the update ran: the ids follow (kept landmark j of the frame is f->ids[b][j])
        for (int b = 0; b < B; ++b) {
            if (!f->hGate[b] || !f->gate.active[b]) continue;
            std::vector<int> nid;
            const int nK = f->gate.nKept[b];
            for (int j = 0; j < int(f->ids[b].size()); ++j)
                if (j >= nK || !(f->hChord[(size_t)b * cap + j] > f->set.outlierThreshold)) nid.push_back(f->ids[b][j]);
            f->ids[b] = nid;
        }
        // (flag 2: the outliers left the filter with so few landmarks that k_edit switched the queued update off -- it runs now, shaped for
        // the count the host knows by now; nothing else of the frame is repeated)
        bool deferred = false;
        int Nmax = 0;
        for (int b = 0; b < B; ++b) {
            const bool d = f->hGate[b] == 2 && f->gate.active[b] && !f->ids[b].empty();
            f->hGate[b] = d ? 1 : 0;
            if (d) {
                deferred = true;
                Nmax = std::max(Nmax, int(f->ids[b].size()));
            }
        }
        if (!deferred) return EQF_OK;
        hipLaunchKernelGGL(k_set_update_ok, dim3((B + 63) / 64), dim3(64), 0, f->stream, f->g[f->pG], f->hGateDev, B);
        HIPC(hipEventRecord(f->evMask, f->stream));
        f->maskPending = true;
        return launchUpdate(f, f->gate.bearings, f->gate.bearStride, f->dPerm, Nmax);
    }
    bool any = false;
    std::vector<char> act(B, 0);
    for (int b = 0; b < B; ++b) {
        const bool hit = f->hGate[b] && f->gate.active[b];
        f->hGate[b] = hit ? 1 : 0;  // the mask k_set_update_ok reads: only the flagged filters take part in the redo
        if (hit) {
            any = true;
            act[b] = 1;
        }
    }
    if (!any) return EQF_OK;
    hipLaunchKernelGGL(k_set_update_ok, dim3((B + 63) / 64), dim3(64), 0, f->stream, f->g[f->pG], f->hGateDev, B);
    HIPC(hipEventRecord(f->evMask, f->stream));
    f->maskPending = true;
    // The flagged filters: their update did not run, their new landmarks of that frame were appended.  One compaction takes out the
    // outliers -- the probe left every chord in pinned memory: no second probe, no readback -- and the appended landmarks (the redo
    // initialises them again, at the median depth of what is left: the reference's order, VIOFilter.cpp:429-443 then :345-391).
    std::vector<std::vector<int>> keep(B);
    std::vector<std::vector<char>> gated(B);
    for (int b = 0; b < B; ++b) {
        const int n = int(f->ids[b].size());
        gated[b].assign(f->gate.nb[b], 0);
        if (!act[b]) {
            keep[b].resize(n);
            for (int i = 0; i < n; ++i) keep[b][i] = i;
            continue;
        }
        const int nOld = std::min(n, f->gate.nOld[b]);
        const int* mi = f->gate.ids[b].data();
        for (int i = 0; i < nOld; ++i) {
            if (f->hChord[(size_t)b * cap + i] > f->set.outlierThreshold) {
                const int* it = std::lower_bound(mi, mi + f->gate.nb[b], f->ids[b][i]);
                gated[b][int(it - mi)] = 1;
            } else {
                keep[b].push_back(i);
            }
        }
    }
    int rc = compact(f, keep);
    if (rc) return rc;
    for (int b = 0; b < B; ++b) {
        std::vector<int> nid;
        for (int o : keep[b]) nid.push_back(f->ids[b][o]);
        f->ids[b] = nid;
    }
    std::vector<const int*> mids(B);
    for (int b = 0; b < B; ++b) mids[b] = f->gate.ids[b].data();
    return visionCore(f, mids, f->gate.nb, f->gate.bearings, f->gate.bearStride, act, nullptr, &gated);
}

void freeAll(eqf_filter* f) {
    if (!f) return;
    for (int p = 0; p < 2; ++p) {
        hipFree(f->Sigma[p]);
        hipFree(f->g[p]);
        hipFree(f->Q[p]);
    }
    for (void* p : {(void*)f->p0, (void*)f->lmc, (void*)f->SA, (void*)f->SL, (void*)f->YW, (void*)f->YO, (void*)f->EA, (void*)f->EL, (void*)f->ZW,
             (void*)f->ZO, (void*)f->dbgDelta, (void*)f->dbgGamma, (void*)f->dbgGammaTot, (void*)f->red, (void*)f->errflag, (void*)f->dMap,
             (void*)f->dPerm, (void*)f->dChord, (void*)f->dDepth2, (void*)f->dDepthSel, (void*)f->dScratch, (void*)f->dMeas,
             (void*)f->dOut, (void*)f->dRing, (void*)f->sImu, (void*)f->sVis, (void*)f->sBear, f->dF, f->dG, f->dBn, f->dBlk, (void*)f->dBlkCommon, f->dColRec, f->dRowRec, (void*)f->dSteps, (void*)f->dFlags, (void*)f->dReadyA, (void*)f->dReadyY, (void*)f->dResCounters, (void*)f->dTicket, (void*)f->dStageFlags, (void*)f->dPrepFlags, (void*)f->dBuildFlags, (void*)f->dGammaPart,
             (void*)f->dG11Part, (void*)f->dRoles})
        hipFree(p);
    if (f->hGate) hipHostFree(f->hGate);
    if (f->dMask) hipFree(f->dMask);
    if (f->evGate) hipEventDestroy(f->evGate);
    if (f->evMask) hipEventDestroy(f->evMask);
    stageFree(f->stMap);
    stageFree(f->stPerm);
    stageFree(f->stEdit);
    hipFree(f->dEdit);
    hipFree(f->dEditBar);
    for (void* p : {(void*)f->hChord, (void*)f->hMeas,
             (void*)f->hOut, (void*)f->hRing})
        if (p) hipHostFree(p);
    for (auto& e : f->evRing)
        if (e) hipEventDestroy(e);
    if (f->evMeas) hipEventDestroy(f->evMeas);
    for (auto e : f->evPool) hipEventDestroy(e);
    for (auto& p : f->profPairs) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    if (f->stream) hipStreamDestroy(f->stream);
    delete f;
}

}  // namespace

extern "C" {

#ifdef EQF_PROP_STAMPS
extern "C" int eqf_debug_prop_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(eqf::g_propStamps), sizeof(long long) * 32) == hipSuccess ? 0 : -1;
}
#endif
#ifdef EQF_STEP64_STAMPS
extern "C" int eqf_debug_step64_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(eqf::g_stamps), sizeof(long long) * 64 * 16) == hipSuccess ? 0 : -1;
}
#endif
#ifdef EQF_CHOL_WG_STAMPS
extern "C" int eqf_debug_chol_wg(long long* t, int* info) {
    if (hipMemcpyFromSymbol(t, HIP_SYMBOL(eqf::g_cholWg), sizeof(long long) * 16 * 256 * 2) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(info, HIP_SYMBOL(eqf::g_cholWgInfo), sizeof(int) * 16 * 256 * 4) == hipSuccess ? 0 : -1;
}
#endif
#ifdef EQF_RES_STAMPS
extern "C" int eqf_debug_res_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(eqf::g_resStamps), sizeof(long long) * 2 * 16 * 16) == hipSuccess ? 0 : -1;
}
#endif
#if defined(EQF_RES_STAMPS) && defined(EQF_F64_STAMPS)
extern "C" int eqf_debug_res_f64_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(eqf::g_resF64), sizeof(long long) * 2 * 16 * 128) == hipSuccess ? 0 : -1;
}
#endif
#ifdef EQF_WAIT_STATS
extern "C" int eqf_debug_wait_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(eqf::g_waitStats), sizeof(unsigned long long) * 48) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[48] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(eqf::g_waitStats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#ifdef EQF_PREP_STAMPS
extern "C" int eqf_debug_prep_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(eqf::g_prepStamps), sizeof(long long) * 1024) == hipSuccess ? 0 : -1;
}
#endif
#ifdef EQF_BURST_STAMPS
extern "C" int eqf_debug_ring_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(eqf::g_ringStamps), sizeof(long long) * 256) == hipSuccess ? 0 : -1;
}
extern "C" int eqf_debug_burst_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(eqf::g_burstStamps), sizeof(long long) * 8 * 20 * 4) == hipSuccess ? 0 : -1;
}
#endif
const char* eqf_version(void) { return "eqf_vio_amd 0.1 (gfx950)"; }

// sha256 of the sources this library was built from (csrc/Makefile: eqf_build_id.inc)
const char* eqf_build_info(void) {
    return "src_sha256="
#include "eqf_build_id.inc"
        ;
}

void eqf_settings_default(eqf_settings* s) {  // VIOFilterSettings.h:29-50
    std::memset(s, 0, sizeof(*s));
    s->biasOmegaProcessVariance = 0.001;
    s->biasAccelProcessVariance = 0.001;
    s->gravityProcessVariance = 0.001;
    s->velocityProcessVariance = 0.001;
    s->pointProcessVariance = 0.001;
    s->velOmegaVariance = 0.1;
    s->velAccelVariance = 0.1;
    s->measurementVariance = 0.1;
    s->initialGravityVariance = 1.0;
    s->initialVelocityVariance = 1.0;
    s->initialPointVariance = 1.0;
    s->initialBiasOmegaVariance = 1.0;
    s->initialBiasAccelVariance = 1.0;
    s->initialSceneDepth = 1.0;
    s->outlierThreshold = 0.01;
    s->useInnovationLift = 1;
    s->useDiscreteInnovationLift = 1;
    s->useDiscreteVelocityLift = 1;
    s->fastRiccati = 0;
    s->cameraOffset_q[0] = 1.0;
}

// camera-offset constants, same formulas as on the device (eqf_math.hpp is host+device)
static void setCameraConstants(Params& p) {
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

int eqf_create(const eqf_settings* settings, int capacity_landmarks, int batch, int device, int precision, eqf_filter** out) {
    if (!settings || !out || capacity_landmarks < 1 || batch < 1) return EQF_ERR_INVALID;
    if (precision != EQF_PRECISION_F64 && precision != EQF_PRECISION_F32) return EQF_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return EQF_ERR_NO_DEVICE;
    HIPC(hipSetDevice(device));
    eqf_filter* f = new eqf_filter();
    f->B = batch;
    f->cap = capacity_landmarks;
    f->device = device;
    f->precision = precision;
    f->esz = precision == EQF_PRECISION_F32 ? 4 : 8;
    f->set = *settings;
    Params& p = f->prm;
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
    setCameraConstants(p);

    const int B = batch, cap = f->cap;
    f->nTot = kLm0 + 3 * cap;
    f->ld = roundUp(f->nTot, 16);
    f->sigmaStride = (long long)f->nTot * f->ld;
    const int mpC = roundUp(2 * cap, kSB), nepC = roundUp(eDim(cap), kSB), ycC = roundUp(yCols(cap), kSB);
    f->ldS = mpC; f->ldY = ycC; f->ldE = nepC; f->ldZ = kSB;
    f->strideS = (long long)mpC * mpC; f->strideY = (long long)mpC * ycC;
    f->strideE = (long long)nepC * nepC; f->strideZ = (long long)nepC * kSB;
    int rc = EQF_OK;
    auto chk = [&](int r) { if (r && !rc) rc = r; };
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) rc = EQF_ERR_HIP;
    for (int q = 0; q < 2 && !rc; ++q) {
        if (hipMalloc(&f->Sigma[q], f->esz * f->sigmaStride * B) != hipSuccess) rc = EQF_ERR_HIP;
        chk(dmalloc(&f->g[q], B));
        chk(dmalloc(&f->Q[q], (size_t)5 * cap * B));
    }
    chk(dmalloc(&f->p0, (size_t)3 * cap * B));
    chk(dmalloc(&f->lmc, (size_t)15 * cap * B));
    f->strideDS = std::max<long long>(f->strideS, (long long)(mpC / kSB) * kDRec);
    f->strideDE = std::max<long long>(f->strideE, (long long)(nepC / kSB) * kDRec);
    chk(dmalloc(&f->SA, f->strideS * B)); chk(dmalloc(&f->SL, f->strideDS * B));
    chk(dmalloc(&f->YW, f->strideY * B)); chk(dmalloc(&f->YO, f->strideY * B));
    chk(dmalloc(&f->EA, f->strideE * B)); chk(dmalloc(&f->EL, f->strideDE * B));
    chk(dmalloc(&f->ZW, f->strideZ * B)); chk(dmalloc(&f->ZO, f->strideZ * B));
    chk(dmalloc(&f->dbgDelta, (size_t)2 * cap * B));
    chk(dmalloc(&f->dbgGamma, (size_t)(kLm0 + 3 * cap) * B));
    chk(dmalloc(&f->dbgGammaTot, (size_t)(9 + 3 * cap) * B));
    chk(dmalloc(&f->red, (size_t)256 * B));
    chk(dmalloc(&f->errflag, 1));
    chk(dmalloc(&f->dMap, (size_t)cap * B + B)); if (!rc) f->dNewN = f->dMap + (size_t)cap * B;  /* (one copy brings both) */ chk(dmalloc(&f->dPerm, (size_t)cap * B));
    chk(dmalloc(&f->dChord, (size_t)cap * B)); chk(dmalloc(&f->dDepth2, (size_t)cap * B));
    chk(dmalloc(&f->dScratch, (size_t)kLmRec * cap * B)); chk(dmalloc(&f->dMeas, (size_t)3 * cap * B + ((size_t)2 * cap * B + 4 * B + 1) / 2));  /* (+ k_edit's image behind the bearings: one copy brings both) */
    chk(dmalloc(&f->dOut, (size_t)f->nTot * f->nTot + 16));
    chk(dmalloc(&f->dRing, (size_t)kRing * B));
    if (!rc && hipMalloc(&f->dBlk, f->esz * (size_t)kBlkRec * cap * B) != hipSuccess) rc = EQF_ERR_HIP;
    chk(dmalloc(&f->dBlkCommon, B));
    if (!rc && hipMalloc(&f->dColRec, f->esz * (size_t)kBurstMax * burstRecStep((long long)kColRec * cap) * B) != hipSuccess) rc = EQF_ERR_HIP;
    if (!rc && hipMalloc(&f->dRowRec, f->esz * (size_t)kBurstMax * burstRecStep((long long)kBlkRec * cap) * B) != hipSuccess) rc = EQF_ERR_HIP;
    chk(dmalloc(&f->dSteps, (size_t)kBurstMax * B));
    f->nBuildCap = ((cap + 3) / 4 + 63) / 64 * 64 + 1088;  // stride of a replica: > 4 KB, not a multiple of it
    chk(dmalloc(&f->dBuildFlags, (size_t)f->nBuildCap * kFlagReplicas * B));
    if (!rc && hipMemset(f->dBuildFlags, 0, sizeof(int) * f->nBuildCap * kFlagReplicas * B) != hipSuccess) rc = EQF_ERR_HIP;
    if (const char* e = std::getenv("EQF_BURST_ROWS")) f->burstRows = (std::atoi(e) == 1 || std::atoi(e) == 2 || std::atoi(e) == 4) ? std::atoi(e) : 0;
    if (const char* e = std::getenv("EQF_IMU_BURST")) f->burstMax = std::max(0, std::min(kBurstMax, std::atoi(e)));
    if (const char* e = std::getenv("EQF_SPLIT_PROPAGATE")) f->splitPropagate = std::atoi(e);
    if (const char* e = std::getenv("EQF_CHOL_SPLIT")) f->cholSplit = std::atoi(e);
    if (const char* e = std::getenv("EQF_CHOL_RESIDENT")) f->cholResident = std::atoi(e);
    if (const char* e = std::getenv("EQF_RES_FOLD_PREP")) f->resFoldPrep = std::atoi(e);
    if (const char* e = std::getenv("EQF_BURST_FUSED")) f->burstFused = std::atoi(e);
    {
        hipDeviceProp_t prop;
        if (!rc && hipGetDeviceProperties(&prop, device) != hipSuccess) rc = EQF_ERR_HIP;
        if (!rc) f->numCUs = prop.multiProcessorCount;
        f->nbCap = std::max(mpC, nepC) / kSB;
        f->wtCap = ycC / kSB;
        // the resident kernel's role table holds one workgroup per 64 x 64 tile; its flag / partial-sum buffers are allocated up to 200 k
        // roles in the batch (one filter of N = 4000: 71 k), beyond that the per-column launches run
        const long long maxRoles = (long long)(f->nbCap + 1) * f->nbCap + (long long)f->wtCap * f->nbCap;
        if (!rc && maxRoles * B <= 200000) {  // (N = 4000: 71 k roles)
            chk(dmalloc(&f->dReadyA, (size_t)2 * f->nbCap * f->nbCap * B));
            chk(dmalloc(&f->dReadyY, (size_t)2 * f->nbCap * f->wtCap * B));
            chk(dmalloc(&f->dResCounters, (size_t)4 * B));
            chk(dmalloc(&f->dTicket, (size_t)32 * B));  // (one counter per filter, a 128-byte line each)
            if (!rc && hipMemset(f->dTicket, 0, sizeof(unsigned) * 32 * B) != hipSuccess) rc = EQF_ERR_HIP;
            chk(dmalloc(&f->dStageFlags, (size_t)2 * f->nbCap * 4 * B));
            if (!rc && hipMemset(f->dStageFlags, 0, sizeof(int) * 2 * f->nbCap * 4 * B) != hipSuccess) rc = EQF_ERR_HIP;
            f->nPrepCap = (mpC / 2 + 3) / 1 + nepC / kNB + 8;  // (landmark workgroups carry >= 1 wave each; Z-row workgroups of kNB rows)
            chk(dmalloc(&f->dPrepFlags, (size_t)f->nPrepCap * B));
            if (!rc && hipMemset(f->dPrepFlags, 0, sizeof(int) * f->nPrepCap * B) != hipSuccess) rc = EQF_ERR_HIP;
            chk(dmalloc(&f->dGammaPart, (size_t)f->nbCap * ycC * B));
            chk(dmalloc(&f->dG11Part, (size_t)f->nbCap * 128 * B));
            if (!rc && (hipMemset(f->dReadyA, 0, sizeof(int) * 2 * f->nbCap * f->nbCap * B) != hipSuccess ||
                        hipMemset(f->dReadyY, 0, sizeof(int) * 2 * f->nbCap * f->wtCap * B) != hipSuccess ||
                        hipMemset(f->dResCounters, 0, sizeof(int) * 4 * B) != hipSuccess))
                rc = EQF_ERR_HIP;
        }
    }
    f->flagStride = std::max(mpC, nepC) / kSB + 1;
    chk(dmalloc(&f->dFlags, (size_t)2 * f->flagStride * B));
    if (!rc && hipMemset(f->dFlags, 0, sizeof(int) * 2 * f->flagStride * B) != hipSuccess) rc = EQF_ERR_HIP;
    if (const char* e = std::getenv("EQF_GATE_SPECULATIVE")) f->gateSpeculative = std::atoi(e);
    chk(stageInit(f->stMap, (size_t)cap * B + B)); chk(stageInit(f->stPerm, (size_t)cap * B));
    chk(stageInit(f->stEdit, (size_t)2 * cap * B + 4 * B)); chk(dmalloc(&f->dEdit, (size_t)2 * cap * B + 4 * B)); chk(dmalloc(&f->dEditBar, (size_t)4 * B));
    if (!rc && hipMemset(f->dEditBar, 0, sizeof(int) * 4 * B) != hipSuccess) rc = EQF_ERR_HIP;
    chk(hmalloc(&f->hChord, (size_t)cap * B)); chk(dmalloc(&f->dDepthSel, B));
    chk(hmalloc(&f->hGate, B)); chk(dmalloc(&f->dMask, B));
    if (!rc && hipHostGetDevicePointer(reinterpret_cast<void**>(&f->hGateDev), f->hGate, 0) != hipSuccess) rc = EQF_ERR_HIP;
    if (!rc && hipEventCreateWithFlags(&f->evGate, hipEventDisableTiming) != hipSuccess) rc = EQF_ERR_HIP;
    if (!rc && hipEventCreateWithFlags(&f->evMask, hipEventDisableTiming) != hipSuccess) rc = EQF_ERR_HIP;
    if (!rc && hipHostGetDevicePointer(reinterpret_cast<void**>(&f->hChordDev), f->hChord, 0) != hipSuccess) rc = EQF_ERR_HIP;
    chk(hmalloc(&f->hMeas, (size_t)3 * cap * B + ((size_t)2 * cap * B + 4 * B + 1) / 2)); chk(hmalloc(&f->hOut, (size_t)f->nTot * f->nTot + 16));
    chk(hmalloc(&f->hRing, (size_t)kRing * B));
    if (!rc) {
        for (auto& e : f->evRing)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = EQF_ERR_HIP;
        if (hipEventCreateWithFlags(&f->evMeas, hipEventDisableTiming) != hipSuccess) rc = EQF_ERR_HIP;
    }
    if (!rc) rc = initState(f);
    if (rc) {
        freeAll(f);
        return rc;
    }
    *out = f;
    return EQF_OK;
}

void eqf_destroy(eqf_filter* f) {
    if (!f) return;
    hipSetDevice(f->device);
    if (f->stream) hipStreamSynchronize(f->stream);
    freeAll(f);
}

int eqf_reset(eqf_filter* f) {
    if (!f) return EQF_ERR_INVALID;
    GATE(f);
    HIPC(hipSetDevice(f->device));
    HIPC(hipStreamSynchronize(f->stream));
    return initState(f);
}

int eqf_synchronize(eqf_filter* f) {
    if (!f) return EQF_ERR_INVALID;
    GATE(f);
    HIPC(hipStreamSynchronize(f->stream));
    return EQF_OK;
}

int eqf_process_imu(eqf_filter* f, const double* stamps, const double* omega, const double* accel, int* status) {
    if (!f || !stamps || !omega || !accel) return EQF_ERR_INVALID;
    const bool burst = burstEligible(f, true);
    if (!(burst && f->B == 1 && f->burst.kind != 1)) GATE(f);  // (a queued call settles nothing: it is only recorded)
    HIPC(hipSetDevice(f->device));
    std::vector<ImuRec> recs(f->B);
    for (int b = 0; b < f->B; ++b) {
        recs[b].stamp = stamps[b];
        for (int i = 0; i < 3; ++i) {
            recs[b].w[i] = omega[3 * b + i];
            recs[b].a[i] = accel[3 * b + i];
        }
        recs[b].pad_ = 0;
    }
    if (burst && f->B == 1) {
        // one filter: queue the record; the burst is launched when it is full or when anything else touches the handle
        auto& q = f->burst;
        q.kind = 2;
        q.inl[q.cnt++] = recs[0];
        mirrorStep(f, stamps, true, status);
        if (q.cnt >= std::min(f->burstMax, kBurstMax - 1) || !allDevInit(f)) {
            GATE(f);
        }
        return EQF_OK;
    }
    const ImuRec* dev = nullptr;
    int slot = -1;
    int rc = stageRecs(f, recs.data(), &dev, &slot);
    if (rc) return rc;
    if (burst) {
        rc = launchBurst(f, 1, dev, 0, nullptr, false, nullptr);
        if (!rc) mirrorStep(f, stamps, true, status);
    } else {
        rc = launchPropagate(f, dev, recs[0], stamps, true, !f->set.fastRiccati, status);
    }
    if (rc) return rc;
    return releaseSlot(f, slot);
}

int eqf_process_vision(eqf_filter* f, const double* stamps, const int* nb, const int* ids, const double* bearings, int stride,
    int* status) {
    if (!f || !stamps || !nb || (!ids && stride > 0) || (!bearings && stride > 0)) return EQF_ERR_INVALID;
    HIPC(hipSetDevice(f->device));
    const int B = f->B, cap = f->cap;
    // Argument errors are reported before ANY effect (nothing enqueued, no time advanced).  After removeOldLandmarks the
    // state's ids are a subset of the measurement's and addNewLandmarks appends the rest, so nb <= capacity is the whole
    // capacity condition of the call (the reference grows Sigma on demand instead, VIOFilter.cpp:386).  Ids must be strictly
    // ascending: the reference asserts is_sorted (VIOFilter.cpp:239-240) and a duplicate id would map two bearings to one
    // landmark / append one landmark twice.
    for (int b = 0; b < B; ++b) {
        if (nb[b] < 0 || nb[b] > stride) return EQF_ERR_INVALID;
        if (nb[b] > cap) return EQF_ERR_CAPACITY;
        for (int k = 1; k < nb[b]; ++k)
            if (ids[(size_t)b * stride + k] <= ids[(size_t)b * stride + k - 1]) return EQF_ERR_UNSORTED;
    }
    {
        const int grc = burstEligible(f, false) ? resolveGate(f) : (resolveGate(f) || flushBurst(f));
        if (grc) return grc;
    }
    // integrateUpToTime(measurement.stamp) (VIOFilter.cpp:234)
    std::vector<ImuRec> recs(B);
    for (int b = 0; b < B; ++b) {
        std::memset(&recs[b], 0, sizeof(ImuRec));
        recs[b].stamp = stamps[b];
    }
    const ImuRec* dev = nullptr;
    int slot = -1;
    int rc = stageRecs(f, recs.data(), &dev, &slot);
    if (rc) return rc;
    std::vector<int> st(B, EQF_OK);
    if (burstEligible(f, false)) {
        rc = flushBurstWith(f, true, dev, &recs[0]);
        if (!rc) mirrorStep(f, stamps, false, st.data());
    } else {
        rc = launchPropagate(f, dev, recs[0], stamps, false, true, st.data());
    }
    if (status) std::copy(st.begin(), st.end(), status);
    if (rc) return rc;
    rc = releaseSlot(f, slot);
    if (rc) return rc;
    // bearings -> device
    HIPC(hipEventSynchronize(f->evMeas));
    std::vector<const int*> mids(B);
    std::vector<int> nbv(nb, nb + B);
    std::vector<char> active(B);
    for (int b = 0; b < B; ++b) {
        std::memcpy(f->hMeas + (size_t)b * 3 * cap, bearings + (size_t)b * stride * 3, sizeof(double) * 3 * nb[b]);
        mids[b] = ids + (size_t)b * stride;
        active[b] = st[b] == EQF_OK;
    }
    f->measPending = true;
    rc = visionCore(f, mids, nbv, f->dMeas, (long long)3 * cap, active, st.data());
    if (f->measPending) {  // (visionCore left before it needed them)
        f->measPending = false;
        HIPC(hipMemcpyAsync(f->dMeas, f->hMeas, sizeof(double) * 3 * cap * B, hipMemcpyHostToDevice, f->stream));
        HIPC(hipEventRecord(f->evMeas, f->stream));
    }
    if (status) std::copy(st.begin(), st.end(), status);
    return rc;
}

int eqf_stream_upload(eqf_filter* f, int K, const double* imu, int F, const double* vstamps, int nbear, const int* ids,
    const double* bearings) {
    if (!f || K < 0 || F < 0 || nbear < 0 || nbear > f->cap) return EQF_ERR_INVALID;
    GATE(f);
    HIPC(hipSetDevice(f->device));
    HIPC(hipStreamSynchronize(f->stream));
    const int B = f->B;
    for (int k = 1; k < nbear; ++k)
        if (ids[k] <= ids[k - 1]) return EQF_ERR_UNSORTED;  // strictly ascending, as in eqf_process_vision
    hipFree(f->sImu); hipFree(f->sVis); hipFree(f->sBear);
    f->sImu = nullptr; f->sVis = nullptr; f->sBear = nullptr;
    std::vector<ImuRec> ri((size_t)K * B), rv((size_t)F * B);
    f->hImuStamp.resize((size_t)K * B);
    f->hVisStamp.resize((size_t)F * B);
    for (size_t e = 0; e < (size_t)K * B; ++e) {
        const double* s = imu + 7 * e;
        ri[e].stamp = s[0];
        for (int i = 0; i < 3; ++i) {
            ri[e].w[i] = s[1 + i];
            ri[e].a[i] = s[4 + i];
        }
        ri[e].pad_ = 0;
        f->hImuStamp[e] = s[0];
    }
    for (size_t e = 0; e < (size_t)F * B; ++e) {
        std::memset(&rv[e], 0, sizeof(ImuRec));
        rv[e].stamp = vstamps[e];
        f->hVisStamp[e] = vstamps[e];
    }
    int rc = dmalloc(&f->sImu, ri.size());
    if (!rc) rc = dmalloc(&f->sVis, rv.size());
    if (!rc) rc = dmalloc(&f->sBear, (size_t)F * B * nbear * 3);
    if (rc) return rc;
    HIPC(hipMemcpy(f->sImu, ri.data(), sizeof(ImuRec) * ri.size(), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(f->sVis, rv.data(), sizeof(ImuRec) * rv.size(), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(f->sBear, bearings, sizeof(double) * (size_t)F * B * nbear * 3, hipMemcpyHostToDevice));
    f->sK = K; f->sF = F; f->sNb = nbear;
    f->sIds.assign(ids, ids + nbear);
    return EQF_OK;
}

int eqf_stream_imu(eqf_filter* f, int k) {
    if (!f || k < 0 || k >= f->sK) return EQF_ERR_INVALID;
    if (burstEligible(f, true)) {
        auto& q = f->burst;
        if (q.kind && !(q.kind == 1 && k == q.k0 + q.cnt)) GATE(f);  // not the record after the queued ones: launch those first
        if (!q.kind) {
            q.kind = 1;
            q.k0 = k;
            q.cnt = 0;
        }
        ++q.cnt;
        mirrorStep(f, f->hImuStamp.data() + (size_t)k * f->B, true, nullptr);
        if (q.cnt >= std::min(f->burstMax, kBurstMax - 1) || !allDevInit(f)) {
            GATE(f);
        }
        return EQF_OK;
    }
    GATE(f);
    ImuRec dummy{};
    return launchPropagate(f, f->sImu + (size_t)k * f->B, dummy, f->hImuStamp.data() + (size_t)k * f->B, true, !f->set.fastRiccati, nullptr);
}

int eqf_stream_vision(eqf_filter* f, int fr) {
    if (!f || fr < 0 || fr >= f->sF) return EQF_ERR_INVALID;
    const int B = f->B;
    ImuRec dummy{};
    std::vector<int> st(B, EQF_OK);
    HIPC(hipSetDevice(f->device));
    int rc = resolveGate(f);
    if (rc) return rc;
    if (burstEligible(f, false)) {
        // the queued IMU steps and this call's integrateUpToTime leave as one burst
        rc = flushBurstWith(f, true, f->sVis + (size_t)fr * B, nullptr);
        if (!rc) mirrorStep(f, f->hVisStamp.data() + (size_t)fr * B, false, st.data());
    } else {
        rc = flushBurst(f);
        if (!rc) rc = launchPropagate(f, f->sVis + (size_t)fr * B, dummy, f->hVisStamp.data() + (size_t)fr * B, false, true, st.data());
    }
    if (rc) return rc;
    std::vector<const int*> mids(B, f->sIds.data());
    std::vector<int> nbv(B, f->sNb);
    std::vector<char> active(B);
    for (int b = 0; b < B; ++b) active[b] = st[b] == EQF_OK;
    return visionCore(f, mids, nbv, f->sBear + (size_t)fr * B * f->sNb * 3, (long long)f->sNb * 3, active, st.data());
}

int eqf_get_time(eqf_filter* f, double* t) {
    if (!f || !t) return EQF_ERR_INVALID;
    std::copy(f->curTime.begin(), f->curTime.end(), t);
    return EQF_OK;
}

int eqf_num_landmarks(eqf_filter* f, int b) {
    if (!f || b < 0 || b >= f->B) return EQF_ERR_INVALID;
    if (resolveGate(f)) return EQF_ERR_HIP;
    return int(f->ids[b].size());
}

int eqf_get_ids(eqf_filter* f, int b, int* ids) {
    if (!f || b < 0 || b >= f->B || !ids) return EQF_ERR_INVALID;
    GATE(f);
    std::copy(f->ids[b].begin(), f->ids[b].end(), ids);
    return EQF_OK;
}

int eqf_get_state_estimate(eqf_filter* f, int b, double* pose_q, double* pose_x, double* velocity, double* p) {
    if (!f || b < 0 || b >= f->B) return EQF_ERR_INVALID;
    GATE(f);
    HIPC(hipSetDevice(f->device));
    const int N = int(f->ids[b].size());
    hipLaunchKernelGGL(k_state_estimate, dim3((N + 255) / 256 + 1), dim3(256), 0, f->stream, f->g[f->pG], b, f->p0, f->Q[f->pG], f->cap, f->dOut);
    HIPC(hipMemcpyAsync(f->hOut, f->dOut, sizeof(double) * (10 + 3 * N), hipMemcpyDeviceToHost, f->stream));
    HIPC(hipStreamSynchronize(f->stream));
    if (pose_q) std::copy(f->hOut, f->hOut + 4, pose_q);
    if (pose_x) std::copy(f->hOut + 4, f->hOut + 7, pose_x);
    if (velocity) std::copy(f->hOut + 7, f->hOut + 10, velocity);
    if (p) std::copy(f->hOut + 10, f->hOut + 10 + 3 * N, p);
    return EQF_OK;
}

static int fetchGlob(eqf_filter* f, int b, Glob* g) {
    HIPC(hipSetDevice(f->device));
    HIPC(hipStreamSynchronize(f->stream));
    HIPC(hipMemcpy(g, f->g[f->pG] + b, sizeof(Glob), hipMemcpyDeviceToHost));
    return EQF_OK;
}

int eqf_get_origin(eqf_filter* f, int b, double* pose_q, double* pose_x, double* velocity, double* p) {
    if (!f || b < 0 || b >= f->B) return EQF_ERR_INVALID;
    GATE(f);
    Glob g;
    int rc = fetchGlob(f, b, &g);
    if (rc) return rc;
    if (pose_q) std::copy(g.P0q, g.P0q + 4, pose_q);
    if (pose_x) std::copy(g.P0x, g.P0x + 3, pose_x);
    if (velocity) std::copy(g.v0, g.v0 + 3, velocity);
    if (p) {
        const int N = int(f->ids[b].size()), cap = f->cap;
        std::vector<double> tmp((size_t)3 * cap);
        HIPC(hipMemcpy(tmp.data(), f->p0 + (size_t)b * 3 * cap, sizeof(double) * 3 * cap, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < 3; ++c) p[3 * i + c] = tmp[(size_t)c * cap + i];
    }
    return EQF_OK;
}

int eqf_get_group(eqf_filter* f, int b, double* A_q, double* A_x, double* w, double* Q_q, double* Q_a) {
    if (!f || b < 0 || b >= f->B) return EQF_ERR_INVALID;
    GATE(f);
    Glob g;
    int rc = fetchGlob(f, b, &g);
    if (rc) return rc;
    if (A_q) std::copy(g.Aq, g.Aq + 4, A_q);
    if (A_x) std::copy(g.Ax, g.Ax + 3, A_x);
    if (w) std::copy(g.w, g.w + 3, w);
    if (Q_q || Q_a) {
        const int N = int(f->ids[b].size()), cap = f->cap;
        std::vector<double> tmp((size_t)5 * cap);
        HIPC(hipMemcpy(tmp.data(), f->Q[f->pG] + (size_t)b * 5 * cap, sizeof(double) * 5 * cap, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i) {
            if (Q_q)
                for (int c = 0; c < 4; ++c) Q_q[4 * i + c] = tmp[(size_t)c * cap + i];
            if (Q_a) Q_a[i] = tmp[(size_t)4 * cap + i];
        }
    }
    return EQF_OK;
}

int eqf_get_bias(eqf_filter* f, int b, double* bias6) {
    if (!f || b < 0 || b >= f->B || !bias6) return EQF_ERR_INVALID;
    GATE(f);
    Glob g;
    int rc = fetchGlob(f, b, &g);
    if (rc) return rc;
    std::copy(g.bias, g.bias + 6, bias6);
    return EQF_OK;
}

int eqf_get_sigma(eqf_filter* f, int b, double* dst, int ld) {
    if (!f || b < 0 || b >= f->B || !dst) return EQF_ERR_INVALID;
    GATE(f);
    HIPC(hipSetDevice(f->device));
    const int n = kBase + 3 * int(f->ids[b].size());
    if (ld < n) return EQF_ERR_INVALID;
    const dim3 grid((n + 255) / 256, n);
    if (f->precision == EQF_PRECISION_F32)
        hipLaunchKernelGGL(k_sigma_export<float>, grid, dim3(256), 0, f->stream,
            static_cast<const float*>(f->Sigma[f->pS]) + (long long)b * f->sigmaStride, f->ld, n, f->dOut, n);
    else
        hipLaunchKernelGGL(k_sigma_export<double>, grid, dim3(256), 0, f->stream,
            static_cast<const double*>(f->Sigma[f->pS]) + (long long)b * f->sigmaStride, f->ld, n, f->dOut, n);
    HIPC(hipMemcpyAsync(f->hOut, f->dOut, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, f->stream));
    HIPC(hipStreamSynchronize(f->stream));
    for (int r = 0; r < n; ++r) std::copy(f->hOut + (size_t)r * n, f->hOut + (size_t)(r + 1) * n, dst + (size_t)r * ld);
    return EQF_OK;
}

int eqf_set_sigma(eqf_filter* f, int b, const double* src, int ld) {
    if (!f || b < 0 || b >= f->B || !src) return EQF_ERR_INVALID;
    GATE(f);
    HIPC(hipSetDevice(f->device));
    const int n = kBase + 3 * int(f->ids[b].size());
    if (ld < n) return EQF_ERR_INVALID;
    f->csValid = false;  // (C Sigma / S left by an earlier burst describe the covariance that is being replaced)
    HIPC(hipStreamSynchronize(f->stream));
    for (int r = 0; r < n; ++r) std::copy(src + (size_t)r * ld, src + (size_t)r * ld + n, f->hOut + (size_t)r * n);
    HIPC(hipMemcpyAsync(f->dOut, f->hOut, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, f->stream));
    const dim3 grid((n + 256) / 256, n + 1);
    if (f->precision == EQF_PRECISION_F32)
        hipLaunchKernelGGL(k_sigma_import<float>, grid, dim3(256), 0, f->stream,
            static_cast<float*>(f->Sigma[f->pS]) + (long long)b * f->sigmaStride, f->ld, n, f->dOut, n);
    else
        hipLaunchKernelGGL(k_sigma_import<double>, grid, dim3(256), 0, f->stream,
            static_cast<double*>(f->Sigma[f->pS]) + (long long)b * f->sigmaStride, f->ld, n, f->dOut, n);
    HIPC(hipStreamSynchronize(f->stream));
    return EQF_OK;
}

int eqf_get_integrator(eqf_filter* f, int b, double* currentVelocity6, double* accumulatedVelocity6, double* accumulatedTime,
    int* initialised) {
    if (!f || b < 0 || b >= f->B) return EQF_ERR_INVALID;
    GATE(f);
    Glob g;
    int rc = fetchGlob(f, b, &g);
    if (rc) return rc;
    if (currentVelocity6) std::copy(g.curVel, g.curVel + 6, currentVelocity6);
    if (accumulatedVelocity6) std::copy(g.accVel, g.accVel + 6, accumulatedVelocity6);
    if (accumulatedTime) *accumulatedTime = g.accTime;
    if (initialised) *initialised = g.initialised;
    return EQF_OK;
}

int eqf_set_state(eqf_filter* f, int b, int N, const int* ids, const double* pose_q, const double* pose_x, const double* velocity,
    const double* p0, const double* A_q, const double* A_x, const double* w, const double* Q_q, const double* Q_a, const double* bias6,
    const double* sigma, int ld, double currentTime, const double* currentVelocity6, const double* accumulatedVelocity6,
    double accumulatedTime, int initialised) {
    if (!f || b < 0 || b >= f->B || N < 0 || !pose_q || !pose_x || !velocity || !A_q || !A_x || !w || !bias6 || !sigma) return EQF_ERR_INVALID;
    GATE(f);
    if (N > 0 && (!ids || !p0 || !Q_q || !Q_a)) return EQF_ERR_INVALID;
    if (N > f->cap) return EQF_ERR_CAPACITY;
    if (ld < kBase + 3 * N) return EQF_ERR_INVALID;
    HIPC(hipSetDevice(f->device));
    Glob g;
    int rc = fetchGlob(f, b, &g);
    if (rc) return rc;
    std::copy(pose_q, pose_q + 4, g.P0q);
    std::copy(pose_x, pose_x + 3, g.P0x);
    std::copy(velocity, velocity + 3, g.v0);
    std::copy(A_q, A_q + 4, g.Aq);
    std::copy(A_x, A_x + 3, g.Ax);
    std::copy(w, w + 3, g.w);
    std::copy(bias6, bias6 + 6, g.bias);
    for (int i = 0; i < 6; ++i) {
        g.curVel[i] = currentVelocity6 ? currentVelocity6[i] : 0.0;
        g.accVel[i] = accumulatedVelocity6 ? accumulatedVelocity6[i] : 0.0;
    }
    g.accTime = accumulatedTime;
    g.curTime = currentTime;
    g.initialised = initialised ? 1 : 0;
    g.N = N;
    g.updateOk = 0;
    HIPC(hipMemcpy(f->g[f->pG] + b, &g, sizeof(Glob), hipMemcpyHostToDevice));
    const int cap = f->cap;
    std::vector<double> tp((size_t)3 * cap, 0.0), tq((size_t)5 * cap, 0.0);
    for (int i = 0; i < N; ++i) {
        for (int c = 0; c < 3; ++c) tp[(size_t)c * cap + i] = p0[3 * i + c];
        for (int c = 0; c < 4; ++c) tq[(size_t)c * cap + i] = Q_q[4 * i + c];
        tq[(size_t)4 * cap + i] = Q_a[i];
    }
    HIPC(hipMemcpy(f->p0 + (size_t)b * 3 * cap, tp.data(), sizeof(double) * 3 * cap, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(f->Q[f->pG] + (size_t)b * 5 * cap, tq.data(), sizeof(double) * 5 * cap, hipMemcpyHostToDevice));
    f->ids[b].assign(ids, ids + N);
    f->curTime[b] = currentTime;
    f->init[b] = initialised ? 1 : 0;
    f->devInit[b] = f->init[b];
    hipLaunchKernelGGL(k_restore_constants, dim3((N + 127) / 128 + 1), dim3(128), 0, f->stream, f->g[f->pG], b, f->p0, f->lmc, cap, f->errflag);
    HIPC(hipGetLastError());
    rc = eqf_set_sigma(f, b, sigma, ld);  // (synchronises)
    if (rc) return rc;
    // Recovery from a hand-off time-out (bit 128 of the sticky flag, include/eqf_vio_amd.h): the launch that timed out wrote no covariance,
    // and once EVERY filter of the handle has been given a state again nothing of it is left -- the bit is cleared (the other bits are
    // the caller's to look at: eqf_reset clears those) and k_edit's barrier counters, which a timed-out launch leaves mid-count, start
    // from zero.  With batch = 1 that is this very call.
    if (f->restoredMark.size() != (size_t)f->B) f->restoredMark.assign(f->B, 0);
    int e = 0;
    HIPC(hipMemcpy(&e, f->errflag, sizeof(int), hipMemcpyDeviceToHost));
    if (!(e & 128)) {
        f->restoredMark.assign(f->B, 0);  // (no time-out to recover from: restores of a healthy handle are not counted towards a later one)
    } else {
        f->restoredMark[b] = 1;
    }
    if ((e & 128) && std::all_of(f->restoredMark.begin(), f->restoredMark.end(), [](char c) { return c != 0; })) {
        f->restoredMark.assign(f->B, 0);
        {
            // (bit 4 with it: a chain that unwinds may have judged a pivot of operands it never received -- a by-product of the time-out)
            e &= ~(128 | 4);
            HIPC(hipMemcpy(f->errflag, &e, sizeof(int), hipMemcpyHostToDevice));
            if (f->dEditBar) HIPC(hipMemset(f->dEditBar, 0, sizeof(int) * 4 * f->B));
        }
    }
    return EQF_OK;
}

int eqf_set_camera_offset(eqf_filter* f, const double* q, const double* x) {
    if (!f || !q || !x) return EQF_ERR_INVALID;
    GATE(f);
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (!(std::fabs(n2 - 1.0) < 1e-6)) return EQF_ERR_INVALID;
    HIPC(hipSetDevice(f->device));
    HIPC(hipStreamSynchronize(f->stream));
    std::copy(q, q + 4, f->prm.camq);
    std::copy(x, x + 3, f->prm.camx);
    std::copy(q, q + 4, f->set.cameraOffset_q);
    std::copy(x, x + 3, f->set.cameraOffset_x);
    setCameraConstants(f->prm);
    return EQF_OK;
}

int eqf_get_last_update(eqf_filter* f, int b, double* delta, double* gamma, double* Gamma) {
    if (!f || b < 0 || b >= f->B) return EQF_ERR_INVALID;
    GATE(f);
    HIPC(hipSetDevice(f->device));
    HIPC(hipStreamSynchronize(f->stream));
    const int N = int(f->ids[b].size()), cap = f->cap;
    if (delta) HIPC(hipMemcpy(delta, f->dbgDelta + (size_t)b * 2 * cap, sizeof(double) * 2 * N, hipMemcpyDeviceToHost));
    if (gamma) {
        std::vector<double> tmp(kLm0 + 3 * N);
        HIPC(hipMemcpy(tmp.data(), f->dbgGamma + (size_t)b * (kLm0 + 3 * cap), sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < kBase; ++i) gamma[i] = tmp[i];
        for (int i = 0; i < 3 * N; ++i) gamma[kBase + i] = tmp[kLm0 + i];
    }
    if (Gamma) HIPC(hipMemcpy(Gamma, f->dbgGammaTot + (size_t)b * (9 + 3 * cap), sizeof(double) * (9 + 3 * N), hipMemcpyDeviceToHost));
    return EQF_OK;
}

int eqf_debug_launch_shape(eqf_filter* f, int* shape8) {
    if (!f || !shape8) return EQF_ERR_INVALID;
    std::copy(f->lastBurstShape, f->lastBurstShape + 8, shape8);
    return EQF_OK;
}

int eqf_debug_get_blocks(eqf_filter* f, int b, double* common, double* rec, double* c0) {
    if (!f || b < 0 || b >= f->B) return EQF_ERR_INVALID;
    if (f->precision != EQF_PRECISION_F64) return EQF_ERR_UNSUPPORTED;
    GATE(f);
    HIPC(hipStreamSynchronize(f->stream));
    const int N = int(f->ids[b].size()), cap = f->cap;
    if (common) {
        CommonLds c;
        HIPC(hipMemcpy(&c, f->dBlkCommon + b, sizeof(CommonLds), hipMemcpyDeviceToHost));
        common[0] = c.T;
        std::copy(c.Bg, c.Bg + 6, common + 1);
        std::copy(c.Bvw, c.Bvw + 9, common + 7);
        std::copy(c.RA, c.RA + 9, common + 16);
        std::copy(c.Avg, c.Avg + 6, common + 25);
    }
    if (rec && N > 0) {
        std::vector<double> tmp((size_t)kBlkRec * N);
        HIPC(hipMemcpy(tmp.data(), static_cast<const double*>(f->dBlk) + (size_t)b * cap * kBlkRec, sizeof(double) * tmp.size(),
            hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i) std::copy(&tmp[(size_t)i * kBlkRec], &tmp[(size_t)i * kBlkRec] + 27, rec + (size_t)27 * i);
    }
    if (c0 && N > 0) {
        std::vector<double> tmp((size_t)15 * cap);
        HIPC(hipMemcpy(tmp.data(), f->lmc + (size_t)b * 15 * cap, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i)
            for (int q = 0; q < 6; ++q) c0[(size_t)6 * i + q] = tmp[(size_t)q * cap + i];
    }
    return EQF_OK;
}

int eqf_device_error(eqf_filter* f) {
    if (!f) return EQF_ERR_INVALID;
    if (resolveGate(f) || flushBurst(f)) return EQF_ERR_HIP;
    if (hipSetDevice(f->device) != hipSuccess) return EQF_ERR_HIP;
    if (hipStreamSynchronize(f->stream) != hipSuccess) return EQF_ERR_HIP;
    int e = 0;
    if (hipMemcpy(&e, f->errflag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return EQF_ERR_HIP;
    return e;
}

int eqf_debug_drop_role(eqf_filter* f, int kind, int role, int R, int C) {
    if (!f || kind > 1 || (kind >= 0 && (role < 0 || role > 2))) return EQF_ERR_INVALID;
    GATE(f);
    f->dropRole[0] = kind; f->dropRole[1] = role; f->dropRole[2] = R; f->dropRole[3] = C;
    f->rolesN = -1;  // the role table is rebuilt by the next update
    return EQF_OK;
}

int eqf_debug_option(eqf_filter* f, const char* name, int value) {
    if (!f || !name) return EQF_ERR_INVALID;
    GATE(f);
    if (!std::strcmp(name, "e_sigma_min_percu_x10")) {
        f->eSigmaMinPerCU = value / 10.0;
        return EQF_OK;
    }
    if (!std::strcmp(name, "burst_fused_max_x10")) {
        f->fusedMaxPerCU = value / 10.0;
        return EQF_OK;
    }
    if (!std::strcmp(name, "ring_ahead2")) {
        f->ringAhead2 = value ? 1 : 0;
        return EQF_OK;
    }
    if (!std::strcmp(name, "burst_lm")) {  // landmarks per builder workgroup: 0 = by launch size, 4 / 8 / 16
        if (value != 0 && value != 4 && value != 8 && value != 16) return EQF_ERR_INVALID;
        f->burstLm = value;
        return EQF_OK;
    }
    if (!std::strcmp(name, "burst_rows")) {  // row landmarks per wavefront of the block kernel: 0 = by launch size, 1 / 2 / 4
        if (value != 0 && value != 1 && value != 2 && value != 4) return EQF_ERR_INVALID;
        f->burstRows = value;
        return EQF_OK;
    }
    if (!std::strcmp(name, "res_tickets")) {
        if (value < 0 || value > 2) return EQF_ERR_INVALID;
        f->resTickets = value;
        return EQF_OK;
    }
    if (!std::strcmp(name, "device_edit")) {
        f->deviceEdit = value ? 1 : 0;
        return EQF_OK;
    }
    if (!std::strcmp(name, "cs_in_burst")) {
        f->csInBurst = value;
        f->csValid = false;
        return EQF_OK;
    }
    return EQF_ERR_INVALID;
}

int eqf_set_imu_burst(eqf_filter* f, int max_steps) {
    if (!f || max_steps < 0) return EQF_ERR_INVALID;
    GATE(f);
    f->burstMax = std::min(max_steps, kBurstMax);
    return EQF_OK;
}

int eqf_set_dense_propagate(eqf_filter* f, int on) {
    if (!f) return EQF_ERR_INVALID;
    GATE(f);
    HIPC(hipSetDevice(f->device));
    if (on && !f->dF) {
        const size_t bytes = f->esz * f->sigmaStride * f->B;
        HIPC(hipMalloc(&f->dF, bytes));
        HIPC(hipMalloc(&f->dG, bytes));
        HIPC(hipMalloc(&f->dBn, f->esz * (size_t)f->nTot * 6 * f->B));
        HIPC(hipMemset(f->dF, 0, bytes));
        HIPC(hipMemset(f->dG, 0, bytes));
    }
    f->densePropagate = on ? 1 : 0;
    f->csValid = false;
    return EQF_OK;
}

int eqf_profile_enable(eqf_filter* f, int on) {
    if (!f) return EQF_ERR_INVALID;
    GATE(f);
    if (f->prof && !on) {
        int rc = profDrain(f);
        if (rc) return rc;
    }
    f->prof = on != 0;
    if (on) {
        // calibrate the bracket itself: two event records with nothing in between still measure a few microseconds,
        // which would otherwise be charged to every kernel
        std::fill(std::begin(f->profCount), std::end(f->profCount), 0);
        std::fill(std::begin(f->profMs), std::end(f->profMs), 0.0);
        for (auto& v : f->profSamples) v.clear();
        f->profOverheadMs = 0.0;
        // (median of 64 empty brackets: a single hiccup must not be charged to every kernel)
        for (int i = 0; i < 64; ++i) {
            int rc = profiled(f, EQF_PROF_CHURN, [] {});
            if (rc) return rc;
        }
        HIPC(hipStreamSynchronize(f->stream));
        std::vector<float> el;
        for (auto& p : f->profPairs) {
            float ms = 0;
            HIPC(hipEventElapsedTime(&ms, p.a, p.b));
            el.push_back(ms);
            f->evPool.push_back(p.a);
            f->evPool.push_back(p.b);
        }
        f->profPairs.clear();
        std::sort(el.begin(), el.end());
        f->profOverheadMs = el.empty() ? 0.0 : el[el.size() / 2];
    }
    return EQF_OK;
}

int eqf_profile_get(eqf_filter* f, int cls, long long* launches, double* total_ms) {
    if (!f || cls < 0 || cls >= EQF_PROF_CLASSES) return EQF_ERR_INVALID;
    GATE(f);
    int rc = profDrain(f);
    if (rc) return rc;
    if (launches) *launches = f->profCount[cls];
    if (total_ms) {
        // An event bracket measures kernel time PLUS whatever the stream waited for the host between the two records (the
        // profiled pass triples the API calls per launch, so a small problem is host-bound in it: brackets of one and the
        // same launch then read 13 us or 40 us by chance).  Launches of one class and one shape key do identical work, so
        // each shape is charged its MEDIAN bracket times its launch count; different shapes (chain step 0 vs step 9, the
        // launch that carries the downdate) are never compared with each other.
        std::vector<ProfSample> v = f->profSamples[cls];
        std::sort(v.begin(), v.end(), [](const ProfSample& x, const ProfSample& y) { return x.key != y.key ? x.key < y.key : x.ms < y.ms; });
        double sum = 0.0;
        for (size_t i = 0; i < v.size();) {
            size_t j = i;
            while (j < v.size() && v[j].key == v[i].key) ++j;
            sum += double(v[i + (j - i) / 2].ms) * double(j - i);
            i = j;
        }
        *total_ms = std::max(0.0, sum - f->profOverheadMs * f->profCount[cls]);
    }
    return EQF_OK;
}

const char* eqf_profile_class_name(int cls) {
    static const char* names[EQF_PROF_CLASSES] = {"k_propagate", "k_update_prep", "k_chol_step", "k_update_reduce", "k_update_finish",
        "k_downdate", "churn", "k_dense_riccati", "k_imu_burst", "k_chol_step_dd", "k_chol_resident"};
    return (cls >= 0 && cls < EQF_PROF_CLASSES) ? names[cls] : "?";
}

}  // extern "C"
