// The HOST LOOP of the 2-D block-partitioned filter (BASELINE configs[4]) behind the C ABI: eqf_tf_* (include/eqf_vio_amd.h).
//
// One eqf_tf per rank of a Pr x Pc process grid is VIOFilter (eqf_vio/include/eqf_vio/VIOFilter.h:41-88) for ONE filter whose Sigma is
// partitioned over the grid: processIMUData / processVisionData (VIOFilter.cpp:120-131, :232-302), the landmark bookkeeping on slots
// (:211-230, :345-443), the getters.  Everything a rank does on its own GPU goes through the per-rank entry points of eqf_tiled.hip
// (eqf_tiled_*: replicated O(N) state, base panel, local blocks) and the dense tile kernels (eqf_tile_*); what the ranks exchange -- the
// solved block rows of the two distributed Cholesky factorisations of an update, a few small gathers -- goes through ONE callback,
// eqf_tf_comm::bcast (device pointers + the HIP stream the transfer must be ordered on): a C++ host hands in RCCL (ncclBroadcast on its
// row / column communicators), the Python glue of this repository hands in torch.distributed (eqf_vio_amd/tiled.py).  No collective
// library is linked here.
//
// The schedule is the one tests/tiled_reference.py spells out in Python (round 3/4: eqf_vio_amd/tiled.py): right-looking blocked
// Cholesky by block ROWS, SUMMA-restricted broadcasts (a rank receives 1/Pr + 1/Pc of every block row), look-ahead factorisation of the
// next diagonal block on a CU-masked side stream, the two factorisations of an update side by side on two stream pairs, the covariance
// downdate as ONE product per update.  Same kernels in the same order on the same operands: bitwise the Python reference's results
// (tests/test_gpu_tiled.py).  Until round 5 that loop lived in Python and issued several hundred ctypes calls per update.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <tuple>
#include <unordered_set>
#include <vector>

#include "../../include/eqf_vio_amd_debug.h"  // (the public header + the test / measurement hooks this library also exports)

#define HIPC(expr)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "eqf_vio_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return EQF_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)
#define RC(expr)                 \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ < 0) return rc_; \
    } while (0)

extern "C" __attribute__((visibility("hidden"))) int eqf_tiled_bearings_consumed(eqf_tiled* t, void* stream);  // csrc/eqf_tiled.hip

namespace {
constexpr int kNarrowS = 18;  // (C Sigma)_Ib (11) | delta | V (6)
constexpr int kNarrowE = 11;  // Z_P (6) | E_top (5)
constexpr int kDRec = 64 * 64 + 4 * 16 * 16;
constexpr int kBurstMax = 16;
constexpr int kTrsmSplit = 6;
constexpr int kPhases = 7;
const char* const kPhaseNames[kPhases] = {"propagate", "churn", "prep", "chain_S", "chain_E", "downdate", "finish"};

// out[c][r] = in[r][c] for an (rows x cols) view: the transposed operand of a split block-row solve (trsmLeft)
__global__ __launch_bounds__(256) void k_tf_transpose(double* out, int ldo, const double* in, int ldi, int rows, int cols) {
    __shared__ double tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(long long)(r0 + i) * ldi + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(long long)(c0 + i) * ldo + r0 + tx] = tile[tx][i];
}
// out (n x n, leading dimension ld) <- I: the right-hand side of a diagonal factor's inverse (invertFactor)
__global__ void k_tf_eye(double* out, int ld, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * n) out[(long long)(i / n) * ld + i % n] = (i / n == i % n) ? 1.0 : 0.0;
}
__global__ void k_tf_add(double* out, const double* a, const double* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

struct View {
    double* p = nullptr;
    int ld = 0, r = 0, c = 0;
    View sub(int r0, int r1, int c0, int c1) const { return View{p + (long long)r0 * ld + c0, ld, r1 - r0, c1 - c0}; }
    View rows(int r0, int r1) const { return sub(r0, r1, 0, c); }
    View cols(int c0, int c1) const { return sub(0, r, c0, c1); }
    bool empty() const { return r <= 0 || c <= 0 || !p; }
};
View flat(double* p, int r, int c) { return View{p, c, r, c}; }

// Geometry of the partition: N landmark slots in blocks of bl, block I on process row I mod Pr, block J on process column J mod Pc.
struct Geo {
    int N = 0, bl = 1, Pr = 1, Pc = 1, pr = 0, pc = 0, nb = 0, nlr = 0, nlc = 0;
    std::vector<int> rowBlocks, colBlocks, rowMap, colMap;
    Geo() {}
    Geo(int N_, int bl_, int Pr_, int Pc_, int pr_, int pc_) : N(N_), bl(bl_), Pr(Pr_), Pc(Pc_), pr(pr_), pc(pc_) {
        nb = (N + bl - 1) / bl;
        for (int b = pr; b < nb; b += Pr) rowBlocks.push_back(b);
        for (int b = pc; b < nb; b += Pc) colBlocks.push_back(b);
        for (int b : rowBlocks)
            for (int i = 0; i < blockSize(b); ++i) rowMap.push_back(b * bl + i);
        for (int b : colBlocks)
            for (int i = 0; i < blockSize(b); ++i) colMap.push_back(b * bl + i);
        nlr = (int)rowMap.size();
        nlc = (int)colMap.size();
    }
    int blockSize(int b) const { return std::min(bl, N - b * bl); }
    int ncolsOf(int c) const {  // landmarks in the local columns of process column c
        int s = 0;
        for (int b = c; b < nb; b += Pc) s += blockSize(b);
        return s;
    }
    int nrowsOf(int r) const {
        int s = 0;
        for (int b = r; b < nb; b += Pr) s += blockSize(b);
        return s;
    }
    static int blocksUpto(int k, int p, int P) { return k >= p ? (k - p) / P + 1 : 0; }  // blocks b <= k with b mod P == p
};

struct Piece {
    View v;
    int jl0 = 0;
};
typedef std::map<int, Piece> Contributions;

struct ChainBufs {  // everything double-buffered (k & 1): block row k + 1 is solved and broadcast while the products of block row k still read k's
    std::map<int, double*> buf[2];  // per process column of my row (and my own): a solved block row piece
    double* pack[2] = {nullptr, nullptr};
    double* aopA[2] = {nullptr, nullptr};
};
}  // namespace

struct eqf_tf {
    int cap = 0, bl = 1, Pr = 1, Pc = 1, rank = 0, world = 1, pr = 0, pc = 0, device = 0;
    eqf_settings set{};
    eqf_tf_comm comm{};
    bool haveComm = false;
    eqf_tiled* t = nullptr;
    hipStream_t sMain = nullptr, sSide = nullptr, sAux = nullptr, sAuxSide = nullptr, cur = nullptr;
    hipStream_t sPanel[2] = {nullptr, nullptr};  // per chain: the NEXT block row's solve + exchanges, next to the current block row's products
    hipStream_t sComm = nullptr;                 // every exchange, in program order: one communicator is enough
    int panelAhead = 0;  // set at creation: 1 on more than one rank (there are exchanges to hide), 0 on a 1 x 1 grid -- see eqf_tf_create
    bool ownStreams = false;
    int reserve = 24;
    int* info = nullptr;  // device: or-ed with 1 when a pivot of a diagonal block was not positive
    // options
    int lookahead = 1, overlapChains = 1, burst = 1, checkEvery = 1, framesSinceCheck = 0, profiling = 0, graphs = 0;
    // "downdate_slices" (round 6): 0 = the downdate Sigma - Y^T Y on the fp64 matrix cores (default, parity grade); 5 / 6 / 7 = on the INTEGER
    // matrix pipe from that many 7-bit slices of Y's columns, exact accumulation (eqf_tile_downdate_i8): 34 / 41 / 48 bits of every entry
    // relative to its column's largest -- Sigma within 1e-4 of the fp64 path from SIX slices on: measured 2e-6 .. 6e-5 at N = 200 .. 4000, five
    // miss it (1.4e-4 .. 9e-4: profiles/r06_i8_downdate_error.txt, r06_slice_precision_study_2s_with_pairs.txt)
    int ddSlices = 0;
    void* i8Work = nullptr;
    size_t i8WorkBytes = 0;
    // "chain_slices" (round 6): the same for the trailing products of the two factorisations (every block row's U_k^T U_k and U_k^T Y_k; the
    // diagonal blocks' look-ahead products, the solves and the factors stay fp64).  A factorisation forgives more than the downdate -- the
    // products are subtracted from S and Sigma_e, not from Sigma, and K = Sigma C^T S^-1 averages the error over 2 N rows: FIVE slices keep Sigma
    // to 1e-8 and the pose to 3e-9 of the fp64 path over the bench stream (scripts/slice_precision_study_chain.py,
    // profiles/r06_slice_precision_study_chain.txt).  One workspace per factorisation (they run next to each other).
    int trsmLeaf = 0;  // > 0: block-row solves split recursively down to this many 64-row blocks (0: one split); see trsmLeft
    // "downdate_early" (round 6; percent, with overlapping chains and the fp64 downdate): the downdate Sigma -= Y^T Y is a sum over the S-chain's
    // block rows, and block row k's share Y_k^T Y_k can be subtracted as soon as that block row is solved.  The S-chain's stream carries the
    // S-chain AND the downdate (39 + 22 ms at N = 4000), the E-chain's stream 41 ms: the shares of the first `downdate_early` percent of the
    // block rows are issued on the E-chain's stream, one block row behind the S-chain, the rest stays one product behind the S-chain.
    int ddEarly = 0;
    // "solve_inverse" (round 6): a block row's solve R <- L_kk^-1 R as ONE product with the explicit inverse of the diagonal factor.  The strip
    // solves of eqf_tile_trsm are latency chains that hold a CU each -- 29 ms of kernel time per update at N = 4000 for 0.12 TFLOP, at 4 TFLOP/s,
    // on CUs the other factorisation's products then do not get; the product costs twice the flops at 55.  L_kk^-T comes from the same strip
    // kernel on an identity right-hand side (bk / 64 workgroups), behind the factor: for the look-ahead blocks on the reserved CUs, in the
    // shadow of the previous block row's products.  bk <= 750: the inverse of a Cholesky factor block, error ~ cond(L_kk) eps.
    int solveInverse = 0;
    int chainSlices = 0;
    void* chainWork[2] = {nullptr, nullptr};
    size_t chainWorkBytes[2] = {0, 0};
    // one rank: the launch sequence of an update is fixed for a given number of slots and buffer parity -- it CAN be captured once as a hipGraph
    // (the second update of that shape: the first allocates its scratch operands) and replayed: one hipGraphLaunch instead of ~2000 launches.
    // OFF by default (option "graphs" / EQF_TILED_GRAPHS=1): measured on the MI355X with ROCm 7.2 (profiles/r05_tiled_host_loop.txt) the
    // replay costs the host 0.1 ms per update instead of 4 .. 35 ms -- and the GPU 110 ms per frame instead of 68 at N = 4000 (19 instead of
    // 5 at N = 1000): the graph's nodes do not keep the four streams' concurrency nor the CU masks of the look-ahead streams.  The update
    // is bound by the GPU, not by the host, so the plain launches stay; and inside a process that runs on torch's bundled HIP runtime the
    // capture of CU-masked streams crashed.
    std::map<std::pair<int, int>, hipGraphExec_t> graphExec;
    std::map<std::pair<int, int>, int> graphSeen;
    long long graphLaunches = 0;
    // geometry + storage (allocated once for `cap` slots; the working set is a view of it for the slots in use)
    bool allocated = false, haveGeo = false, symmetric = false;
    Geo full, geo;
    View SllBuf, MBuf, EBuf, Sll, M, E, YcBuf, YrBuf, Yc, Yr, accSBuf, accS, accE, aopW;
    double *G11 = nullptr, *G11sum = nullptr, *GnnBuf = nullptr;
    ChainBufs bufS, bufE;
    std::vector<int> wmaxS, wmaxE;
    std::map<std::tuple<int, int, hipStream_t>, double*> l21t;
    std::vector<hipEvent_t> yReady;  // per block row of the S-chain: Y_k has been copied into the downdate's operands (this update)
    std::map<std::pair<int, int>, std::vector<double*>> padBufs;  // (rows, cols) -> buffers of the small gathers
    std::vector<double*> allocs;
    // landmark bookkeeping (host; identical on every rank): ids in the REFERENCE's order (VIOFilter.cpp:211-230), the slot of each
    bool haveIds = false;
    std::vector<int> ids, slotOf;
    std::vector<char> taken;
    int nslots = 0;
    long long stats[3] = {0, 0, 0};  // removed_old, removed_outliers, added
    // IMU calls waiting for their burst; the filter's time as the queued calls leave it
    struct Rec {
        double stamp, w[3], a[3];
    };
    std::vector<Rec> queue;
    bool mirrorValid = false;
    double mirrorTime = -1.0;
    // events
    std::vector<hipEvent_t> evPool;
    size_t evNext = 0;
    struct PhasePair {
        int phase;
        hipEvent_t a, b;
    };
    std::vector<PhasePair> phasePending;
    double phaseMs[kPhases] = {};
    std::string lastError;
};

namespace {
struct CurStream {  // (the Python reference's `with be.main():` / `with side():`)
    eqf_tf* f;
    hipStream_t prev;
    CurStream(eqf_tf* f_, hipStream_t s) : f(f_), prev(f_->cur) { f->cur = s; }
    ~CurStream() { f->cur = prev; }
};
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

int dalloc(eqf_tf* f, double** p, size_t count, bool zero) {
    HIPC(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(double)));
    if (zero) HIPC(hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(double)));
    f->allocs.push_back(*p);
    return EQF_OK;
}
int dview(eqf_tf* f, View* v, int r, int c, bool zero) {
    double* p = nullptr;
    int rc = dalloc(f, &p, (size_t)r * c, zero);
    if (rc) return rc;
    *v = flat(p, r, c);
    return EQF_OK;
}
int copy2d(eqf_tf* f, View dst, View src) {  // dst <- src (same shape), on the current stream
    if (dst.empty()) return EQF_OK;
    HIPC(hipMemcpy2DAsync(dst.p, sizeof(double) * dst.ld, src.p, sizeof(double) * src.ld, sizeof(double) * dst.c, dst.r, hipMemcpyDeviceToDevice, f->cur));
    return EQF_OK;
}
int zero2d(eqf_tf* f, View v) {
    if (v.empty()) return EQF_OK;
    HIPC(hipMemset2DAsync(v.p, sizeof(double) * v.ld, 0, sizeof(double) * v.c, v.r, f->cur));
    return EQF_OK;
}
hipEvent_t record(eqf_tf* f) {
    if (f->evNext >= f->evPool.size()) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        f->evPool.push_back(e);
    }
    hipEvent_t e = f->evPool[f->evNext++];
    hipEventRecord(e, f->cur);
    return e;
}
void wait(eqf_tf* f, hipEvent_t e) {
    if (e) hipStreamWaitEvent(f->cur, e, 0);
}
struct Phase {  // event bracket around a phase of a call, on the current stream (eqf_tf_set_profiling)
    eqf_tf* f;
    int ph;
    hipEvent_t a = nullptr, b = nullptr;
    Phase(eqf_tf* f_, int ph_) : f(f_), ph(ph_) { enter(); }
    void enter() {
        if (!f->profiling || a) return;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a, f->cur);
    }
    void exit() {
        if (!a || !b) return;
        hipEventRecord(b, f->cur);
        f->phasePending.push_back({ph, a, b});
        a = b = nullptr;
    }
    ~Phase() { exit(); }
};

// ---- exchanges: contiguous device buffers, root = process column (row group), process row (column group) or rank (everybody)
int bcast(eqf_tf* f, int group, int chain, int root, double* p, size_t count) {
    if (f->world == 1 || count == 0) return EQF_OK;
    if (group == 0 && f->Pc == 1) return EQF_OK;
    if (group == 1 && f->Pr == 1) return EQF_OK;
    if (!f->haveComm || !f->comm.bcast) return EQF_ERR_INVALID;
    // every exchange of the handle runs on ONE stream, in program order (identical on every rank): whatever the chains' streams overlap, the
    // collective library sees one ordered sequence per communicator.  The calling stream hands over and takes back through events.
    if (!f->sComm || f->sComm == f->cur) {
        const int rc = f->comm.bcast(f->comm.ctx, group, chain, root, p, count * sizeof(double), f->cur);
        return rc == 0 ? EQF_OK : EQF_ERR_HIP;
    }
    hipEvent_t before = record(f), after = nullptr;
    int rc;
    {
        CurStream cs(f, f->sComm);
        wait(f, before);
        rc = f->comm.bcast(f->comm.ctx, group, chain, root, p, count * sizeof(double), f->cur);
        after = record(f);
    }
    wait(f, after);
    return rc == 0 ? EQF_OK : EQF_ERR_HIP;
}

// ---- dense tile kernels on views, on the current stream
int potrf(eqf_tf* f, View A, double* drec) { return eqf_tile_potrf(f->device, f->cur, A.p, A.ld, A.r, drec, f->info); }
int gemmTn(eqf_tf* f, View C, View A, View B, double alpha, const int* mask = nullptr) {
    if (C.r <= 0 || C.c <= 0 || A.r <= 0) return EQF_OK;
    static const int none[8] = {0, 0, 0, 1, 0, 0, 1, 0};
    const int* m = mask ? mask : none;
    return eqf_tile_gemm_tn(f->device, f->cur, C.p, C.ld, C.r, C.c, A.p, A.ld, B.p, B.ld, A.r, alpha, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
}
// Tt (n x n) <- L^-T on the current stream: I L^-T by the strip kernel (a workgroup per 64 rows); see solveInverse
int invertFactor(eqf_tf* f, View L, const double* drec, View Tt) {
    const int n = L.r;
    hipLaunchKernelGGL(k_tf_eye, dim3((n * n + 255) / 256), dim3(256), 0, f->cur, Tt.p, Tt.ld, n);
    return eqf_tile_trsm(f->device, f->cur, L.p, L.ld, n, drec, Tt.p, Tt.ld, n, 1);
}
int trsmLaunch(eqf_tf* f, View L, const double* drec, View B) {
    if (B.empty()) return EQF_OK;
    return eqf_tile_trsm(f->device, f->cur, L.p, L.ld, L.r, drec, B.p, B.ld, B.c, 0);
}
// B (n x m) <- L^-1 B.  eqf_tile_trsm is one workgroup per 64-column strip, and a strip is a CHAIN of nb (nb + 1) / 2 block products: its
// time is that chain's latency whatever the width (at N = 4000: 29 ms of kernel time per update, 4 TFLOP/s).  So from kTrsmSplit block rows
// on the solve is split once, [L11 0; L21 L22]: X1 = L11^-1 B1, B2 -= L21 X1 as ONE product on the whole chip (L21 transposed into a
// scratch operand), X2 = L22^-1 B2.  Option "trsm_leaf" > 0 (round 6) splits RECURSIVELY down to that many 64-row blocks (the factor
// transposed once per block row, every L21^T a view of it): measured at N = 4000 it LOSES -- leaf 1 / 2 / 3: 67.3 / 67.4 / 66.2 ms a frame
// against 64.8 (profiles/r06_chain_slices_probe.txt): the leaves' 64 x 64 solves are themselves latency chains and the products of a
// split are short (K = 64 .. 256), while the two factorisations running side by side already fill each other's gaps.  Left as an option.
int trsmSplit(eqf_tf* f, View L, View Lt, const double* drec, View B, int leaf) {
    const int n = L.r, nb = (n + 63) / 64;
    if (nb <= leaf) return trsmLaunch(f, L, drec, B);
    const int h = 64 * (nb / 2);
    RC(trsmSplit(f, L.sub(0, h, 0, h), Lt.sub(0, h, 0, h), drec, B.rows(0, h), leaf));
    RC(gemmTn(f, B.rows(h, n), Lt.sub(0, h, h, n), B.rows(0, h), -1.0));
    return trsmSplit(f, L.sub(h, n, h, n), Lt.sub(h, n, h, n), drec + (size_t)(h / 64) * kDRec, B.rows(h, n), leaf);
}
int trsmLeft(eqf_tf* f, View L, double* drec, View B) {
    const int n = L.r, nb = (n + 63) / 64;
    if (B.empty()) return EQF_OK;
    if (B.c < 256 || (f->trsmLeaf > 0 ? nb <= f->trsmLeaf : nb < kTrsmSplit)) return trsmLaunch(f, L, drec, B);
    const bool once = f->trsmLeaf <= 0;  // (round 5's shape: one split)
    const int h = 64 * (nb / 2);
    const auto key = once ? std::make_tuple(n - h, h, f->cur) : std::make_tuple(n, n, f->cur);
    auto it = f->l21t.find(key);
    if (it == f->l21t.end()) {
        double* p = nullptr;
        RC(dalloc(f, &p, once ? (size_t)h * (n - h) : (size_t)n * n, false));
        it = f->l21t.emplace(key, p).first;
    }
    if (once) {
        RC(trsmLaunch(f, L.sub(0, h, 0, h), drec, B.rows(0, h)));
        const View lt = flat(it->second, h, n - h);
        hipLaunchKernelGGL(k_tf_transpose, dim3((h + 31) / 32, (n - h + 31) / 32), dim3(256), 0, f->cur, lt.p, lt.ld, L.p + (long long)h * L.ld, L.ld, n - h, h);
        RC(gemmTn(f, B.rows(h, n), lt, B.rows(0, h), -1.0));
        return trsmLaunch(f, L.sub(h, n, h, n), drec + (size_t)(h / 64) * kDRec, B.rows(h, n));
    }
    const View Lt = flat(it->second, n, n);
    hipLaunchKernelGGL(k_tf_transpose, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, f->cur, Lt.p, Lt.ld, L.p, L.ld, n, n);
    return trsmSplit(f, L, Lt, drec, B, f->trsmLeaf);
}

// ---- storage
int allocStorage(eqf_tf* f) {
    const int cap = f->cap;
    f->full = Geo(cap, f->bl, f->Pr, f->Pc, f->pr, f->pc);
    const Geo& full = f->full;
    auto r16 = [](int x) { return (x + 15) / 16 * 16; };
    RC(dview(f, &f->SllBuf, std::max(3 * full.nlr, 1), r16(std::max(3 * full.nlc, 1)), true));
    RC(dview(f, &f->MBuf, std::max(2 * full.nlr, 1), r16(5 * full.nlc + kNarrowS), false));
    RC(dview(f, &f->EBuf, std::max(3 * full.nlr, 1), r16(3 * full.nlc + kNarrowE), false));
    RC(dalloc(f, &f->G11, 121, true));
    RC(dalloc(f, &f->G11sum, 121, true));
    RC(dalloc(f, &f->GnnBuf, kNarrowS * kNarrowS, true));
    const int bsmax = 3 * std::min(f->bl, cap);
    std::vector<int> mine;
    for (int c = 0; c < f->Pc; ++c)
        if (c == f->pc || c % f->Pr == f->pr) mine.push_back(c);
    for (int chain = 0; chain < 2; ++chain) {
        ChainBufs& cb = chain ? f->bufE : f->bufS;
        for (int q = 0; q < 2; ++q) {
            for (int c : mine) {
                const int w = chain ? 3 * full.ncolsOf(c) + kNarrowE : 5 * full.ncolsOf(c) + kNarrowS;
                RC(dalloc(f, &cb.buf[q][c], (size_t)bsmax * w, false));
            }
            // [L_kk | its diagonal records | L_kk^-T (option "solve_inverse")]
            RC(dalloc(f, &cb.pack[q], 2 * (size_t)bsmax * bsmax + (size_t)((bsmax + 63) / 64) * kDRec, false));
            RC(dalloc(f, &cb.aopA[q], (size_t)bsmax * std::max(3 * full.nlr, 1), false));
        }
    }
    RC(dview(f, &f->aopW, bsmax, std::max(3 * full.nlr, 1), false));
    // the downdate Sigma_IJ -= sum_k Y_kI^T Y_kJ is ONE product per update: the solved block rows are kept -- the columns of my process
    // column (B operand) and of my row blocks (A operand; the same matrix on a symmetric rank)
    f->symmetric = f->Pr == f->Pc && f->pr == f->pc;
    RC(dview(f, &f->YcBuf, 2 * cap, std::max(3 * full.nlc, 1), false));
    if (f->symmetric) f->YrBuf = f->YcBuf;
    else RC(dview(f, &f->YrBuf, 2 * cap, std::max(3 * full.nlr, 1), false));
    RC(dview(f, &f->accSBuf, kNarrowS, 3 * full.nlc + kNarrowS, true));
    RC(dview(f, &f->accE, kNarrowE, kNarrowE, true));
    f->allocated = true;
    return EQF_OK;
}

// the working set for n >= 1 slots in use: geometry (host + device) and the views of the storage
int setSlots(eqf_tf* f, int n) {
    if (f->haveGeo && f->geo.N == n) return EQF_OK;
    if (!f->allocated) RC(allocStorage(f));
    f->geo = Geo(n, f->bl, f->Pr, f->Pc, f->pr, f->pc);
    f->haveGeo = true;
    const Geo& g = f->geo;
    RC(eqf_tiled_set_geometry(f->t, g.nlr, g.rowMap.data(), g.nlc, g.colMap.data()));
    f->Sll = f->SllBuf.sub(0, 3 * g.nlr, 0, 3 * g.nlc);
    f->M = f->MBuf.sub(0, 2 * g.nlr, 0, 5 * g.nlc + kNarrowS);
    f->E = f->EBuf.sub(0, 3 * g.nlr, 0, 3 * g.nlc + kNarrowE);
    f->wmaxS.assign(f->Pc, 0);
    f->wmaxE.assign(f->Pc, 0);
    for (int c = 0; c < f->Pc; ++c) {
        f->wmaxS[c] = 5 * g.ncolsOf(c) + kNarrowS;
        f->wmaxE[c] = 3 * g.ncolsOf(c) + kNarrowE;
    }
    f->Yc = f->YcBuf.sub(0, 2 * n, 0, 3 * g.nlc);
    f->Yr = f->symmetric ? f->Yc : f->YrBuf.sub(0, 2 * n, 0, 3 * g.nlr);
    f->accS = f->accSBuf.sub(0, kNarrowS, 0, 3 * g.nlc + kNarrowS);
    return EQF_OK;
}

// The A operand of the products of block row k: for each of MY local row blocks (all of them, or the trailing ones i > k) the
// (bk x unit * size) block of the solved block row -- found in the piece of the process column c = i mod Pc, which the rank (pr, c)
// re-broadcast along the process row.  partOf(c, w_c) -> (column offset of the part inside the piece, 1 if the part starts at local block
// jl0_c else 0).  Returns an empty view when there is nothing.
int rowsOperand(eqf_tf* f, const Contributions& con, int unit, const std::function<std::pair<int, int>(int, int)>& partOf, int bk, double* buf,
    bool allBlocks, int k, View* out) {
    const Geo& geo = f->geo;
    const int q = f->Pc / f->Pr, bsF = unit * geo.bl;
    const int il0 = allBlocks ? 0 : Geo::blocksUpto(k, f->pr, f->Pr);
    const int ncol = unit * geo.nlr - il0 * bsF;
    *out = View{};
    if (ncol <= 0) return EQF_OK;
    if (q == 1) {
        // one contributor, c = pr: its local column blocks ARE my local row blocks, in order -> a view, no copy
        const Piece& pc = con.at(f->pr);
        const auto po = partOf(f->pr, pc.v.c);
        const int start = po.first + (po.second ? (il0 - pc.jl0) * bsF : il0 * bsF);
        *out = pc.v.cols(start, start + ncol);
        return EQF_OK;
    }
    const View o = View{buf, unit * geo.nlr, bk, unit * geo.nlr};
    for (int s = 0; s < q; ++s) {
        const int c = f->pr + f->Pr * s;
        const Piece& pc = con.at(c);
        const auto po = partOf(c, pc.v.c);
        // my local row block ilb = s + t q  <->  local column block t of process column c
        for (int ilb = s; ilb < (int)geo.rowBlocks.size(); ilb += q) {
            if (ilb < il0) continue;
            const int t = (ilb - s) / q;
            const int w = unit * geo.blockSize(geo.rowBlocks[ilb]);
            const int src = po.first + (po.second ? (t - pc.jl0) : t) * bsF;
            RC(copy2d(f, o.cols(ilb * bsF, ilb * bsF + w), pc.v.cols(src, src + w)));
        }
    }
    *out = o.cols(il0 * bsF, unit * geo.nlr);
    return EQF_OK;
}

// One of the two distributed factorisations of an update (tests/tiled_reference.py: TiledFilter._chain_steps): blocked right-looking
// Cholesky by block ROWS of the SPD matrix in X[:, :nA] (upper blocks, block size unit * bl, block-cyclic over the grid) with the
// right-hand sides X[:, nA:]; X is consumed.  hook(k, bk, Bop, off, contributions) runs on every rank once block row k is solved:
// Bop[:, off:] holds the right-hand-side part of my process column.  step() does one block row, so that the caller can feed the two
// factorisations to their streams alternately.
struct Chain {
    eqf_tf* f;
    View X;
    int unit, nA, chainId;
    const std::vector<int>* wmax;
    ChainBufs* bufs;
    hipStream_t side, panel;
    hipEvent_t rowReady;  // block row k (the next to be solved) carries every earlier update: at first "the operands are formed"
    std::function<int(int, int, View, int, const Contributions&)> hook;
    int k = 0;
    hipEvent_t ahead = nullptr;               // the look-ahead factor of diagonal block k is in pack[k & 1]
    hipEvent_t updDone[2] = {nullptr, nullptr};  // the products of block row k have read buffer set k & 1
    bool done() const { return k >= f->geo.nb; }
    // One block row.  Two parts on two streams (round 5; until then one stream did both, block row after block row):
    //   PANEL (stream `panel`): the diagonal factor along the process row, my piece of block row k solved, down the process column, along
    //         the process row -- everything the products of block row k need, into buffer set k & 1;
    //   PRODUCTS (the caller's stream): the rank's trailing updates with block row k, the hook (downdate operands, reductions).  The rows of
    //         block k + 1 come FIRST and on their own: as soon as they are updated the panel part of block row k + 1 starts -- its solve
    //         and its exchanges run next to the rest of block row k's products instead of after them.
    int step() {
        const Geo& geo = f->geo;
        const int Pr = f->Pr, Pc = f->Pc, pr = f->pr, pc = f->pc;
        const int bsF = unit * geo.bl, W = X.c, q = k & 1;
        const int prk = k % Pr, pck = k % Pc, bk = unit * geo.blockSize(k), klr = k / Pr, klc = k / Pc;
        const int jl0 = Geo::blocksUpto(k, pc, Pc);
        const int c0 = std::min(jl0 * bsF, nA), width = W - c0;
        const View Bop = flat(bufs->buf[q].at(pc), bk, width);
        Contributions con;
        hipEvent_t panelDone = nullptr;
        {
            // ---- PANEL
            CurStream cs(f, f->panelAhead ? panel : f->cur);
            if (f->panelAhead) {
                wait(f, rowReady);
                wait(f, updDone[q]);  // (buffer set q was read by the products of block row k - 2)
            }
            if (pr == prk) {
                // 1. the diagonal block, L_kk and its records along the process row
                const size_t nrec = (size_t)((bk + 63) / 64) * kDRec;
                double* pack = bufs->pack[q];
                const View Lkk = flat(pack, bk, bk);
                double* drec = pack + (size_t)bk * bk;
                const View Tt = flat(drec + nrec, bk, bk);  // L_kk^-T ("solve_inverse")
                if (pc == pck) {
                    if (ahead) {
                        wait(f, ahead);
                        ahead = nullptr;
                    } else {
                        RC(copy2d(f, Lkk, X.sub(klr * bsF, klr * bsF + bk, klc * bsF, klc * bsF + bk)));
                        RC(potrf(f, Lkk, drec));
                        if (f->solveInverse) RC(invertFactor(f, Lkk, drec, Tt));
                    }
                }
                RC(bcast(f, 0, chainId, pck, pack, (f->solveInverse ? 2 : 1) * (size_t)bk * bk + nrec));
                // 2. my piece of block row k
                const View R = X.sub(klr * bsF, klr * bsF + bk, c0, W);
                if (f->solveInverse) {
                    RC(zero2d(f, Bop));
                    RC(gemmTn(f, Bop, Tt, R, 1.0));  // Bop = (L_kk^-T)^T R
                } else {
                    RC(trsmLeft(f, Lkk, drec, R));
                    RC(copy2d(f, Bop, R));
                }
            }
            // 3. down the process column
            RC(bcast(f, 1, chainId, prk, Bop.p, (size_t)bk * width));
            // 4. along the process row, from the ranks whose column blocks are this process row's row blocks
            for (int c = pr; c < Pc; c += Pr) {
                const int jl0c = Geo::blocksUpto(k, c, Pc);
                const int wc = (*wmax)[c] - std::min(jl0c * bsF, unit * geo.ncolsOf(c));
                const View piece = c == pc ? Bop : flat(bufs->buf[q].at(c), bk, wc);
                RC(bcast(f, 0, chainId, c, piece.p, (size_t)bk * wc));
                con[c] = Piece{piece, jl0c};
            }
            if (f->panelAhead) panelDone = record(f);
        }
        // ---- PRODUCTS: trailing updates of what this rank owns: rows of blocks i > k, columns from block jl0 on
        wait(f, panelDone);
        rowReady = nullptr;
        const int il0 = Geo::blocksUpto(k, pr, Pr);
        if (il0 * bsF < X.r) {
            View Ua;
            RC(rowsOperand(f, con, unit, [](int, int) { return std::make_pair(0, 1); }, bk, bufs->aopA[q], false, k, &Ua));
            const bool ownNextRow = k + 1 < geo.nb && pr == (k + 1) % Pr;  // my first trailing row block IS block row k + 1
            if (f->lookahead && ownNextRow && pc == (k + 1) % Pc) {
                // look-ahead: block (k+1, k+1) is the first trailing block of my rows and of my columns; its factor is ready when the panel
                // part of block row k+1 starts, computed on the side stream (a few reserved CUs) in the shadow of this block row's products
                const int b1 = unit * geo.blockSize(k + 1);
                double* pack1 = bufs->pack[(k + 1) & 1];
                const View L1 = flat(pack1, b1, b1);
                double* drec1 = pack1 + (size_t)b1 * b1;
                RC(copy2d(f, L1, X.sub(il0 * bsF, il0 * bsF + b1, c0, c0 + b1)));
                hipEvent_t ready = record(f);
                {
                    CurStream cs(f, side);
                    wait(f, ready);
                    RC(gemmTn(f, L1, Ua.cols(0, b1), Bop.cols(0, b1), -1.0));
                    RC(potrf(f, L1, drec1));
                    if (f->solveInverse) RC(invertFactor(f, L1, drec1, flat(drec1 + (size_t)((b1 + 63) / 64) * kDRec, b1, b1)));
                    ahead = record(f);
                }
            }
            // the rows of block k + 1 first (when they are mine and somebody is waiting for them), then the rest: the same products on the same
            // elements in two launches instead of one
            const int split = (f->panelAhead && ownNextRow && !f->chainSlices) ? std::min(unit * geo.blockSize(k + 1), X.r - il0 * bsF) : 0;
            if (f->chainSlices > 0) {
                // (the integer pipe: matrix part and right-hand sides in ONE product -- the operands are cut once per block row; Ua is a view of
                // Bop on a grid with Pc == Pr, then one split serves both sides)
                const int m = X.r - il0 * bsF;
                const size_t need = eqf_tile_i8_workspace_bytes(m, width, bk, f->chainSlices, 0);
                if (need > f->chainWorkBytes[chainId]) {
                    HIPC(hipStreamSynchronize(f->cur));
                    if (f->chainWork[chainId]) (void)hipFree(f->chainWork[chainId]);
                    f->chainWork[chainId] = nullptr;
                    f->chainWorkBytes[chainId] = 0;
                    HIPC(hipMalloc(&f->chainWork[chainId], need));
                    f->chainWorkBytes[chainId] = need;
                }
                const int mcols = std::max(nA - c0, 0);
                RC(eqf_tile_gemm_tn_i8(f->device, f->cur, X.p + (size_t)(il0 * bsF) * X.ld + c0, X.ld, m, width, Ua.p, Ua.ld, Bop.p, Bop.ld, bk,
                    f->chainSlices, mcols > 0 ? bsF : 0, mcols > 0 ? bsF : 0, il0, Pr, pr, jl0, Pc, pc, mcols, f->chainWork[chainId],
                    f->chainWorkBytes[chainId]));
                if (f->panelAhead) rowReady = record(f);
            }
            for (int part = 0; part < 2 && !f->chainSlices; ++part) {
                const int r0 = il0 * bsF + (part ? split : 0), r1 = part ? X.r : il0 * bsF + split;
                if (r1 > r0) {
                    const View Ct = X.sub(r0, r1, c0, W), Up = Ua.cols(r0 - il0 * bsF, r1 - il0 * bsF);
                    if (nA - c0 > 0) {
                        const int mask[8] = {bsF, bsF, il0 + (part && split ? 1 : 0), Pr, pr, jl0, Pc, pc};
                        RC(gemmTn(f, Ct.cols(0, nA - c0), Up, Bop.cols(0, nA - c0), -1.0, mask));
                    }
                    RC(gemmTn(f, Ct.cols(nA - c0, width), Up, Bop.cols(nA - c0, width), -1.0));
                }
                if (part == 0 && f->panelAhead) rowReady = record(f);
            }
        } else if (f->panelAhead) {
            rowReady = record(f);
        }
        RC(hook(k, bk, Bop, nA - c0, con));
        if (f->panelAhead) updDone[q] = record(f);
        ++k;
        return EQF_OK;
    }
};

double* padBuffer(eqf_tf* f, int rows, int cols, int idx) {
    auto& v = f->padBufs[{rows, cols}];
    while ((int)v.size() <= idx) {
        double* p = nullptr;
        if (dalloc(f, &p, (size_t)rows * cols, true)) return nullptr;
        v.push_back(p);
    }
    return v[idx];
}

// gamma_L and the base panel's downdate live with the process COLUMNS: gather them along the process row, global landmark order
int gatherColumns(eqf_tf* f, View mine, View* out) {
    const Geo& geo = f->geo;
    const int rows = mine.r;
    int wmax = 0;
    for (int c = 0; c < f->Pc; ++c) wmax = std::max(wmax, 3 * geo.ncolsOf(c));
    wmax = std::max(wmax, 1);
    int wcap = 1;  // (buffers sized for the capacity once, reused by every update)
    for (int c = 0; c < f->Pc; ++c) wcap = std::max(wcap, 3 * f->full.ncolsOf(c));
    double* outp = padBuffer(f, rows, 3 * std::max(f->cap, 1) + 1, 0);
    if (!outp) return EQF_ERR_HIP;
    const View o = View{outp, 3 * geo.N, rows, 3 * geo.N};
    for (int c = 0; c < f->Pc; ++c) {
        double* pp = padBuffer(f, rows, wcap, c);
        if (!pp) return EQF_ERR_HIP;
        const View pad = flat(pp, rows, wmax);
        if (c == f->pc) {
            RC(zero2d(f, pad));
            RC(copy2d(f, pad.cols(0, mine.c), mine));
        }
        RC(bcast(f, 0, 0, c, pad.p, (size_t)rows * wmax));
        int off = 0;
        for (int b = c; b < geo.nb; b += f->Pc) {
            const int w = 3 * geo.blockSize(b);
            RC(copy2d(f, o.cols(3 * b * geo.bl, 3 * b * geo.bl + w), pad.cols(off, off + w)));
            off += w;
        }
    }
    *out = o;
    return EQF_OK;
}

int flushQueue(eqf_tf* f, bool withVision, double visionStamp, int* visionStatus) {
    if (visionStatus) *visionStatus = 0;
    if (f->queue.empty() && !withVision) return EQF_OK;
    const int K = (int)f->queue.size() + (withVision ? 1 : 0);
    double stamps[kBurstMax], w[kBurstMax * 3], a[kBurstMax * 3];
    std::memset(w, 0, sizeof(w));
    std::memset(a, 0, sizeof(a));
    for (size_t k = 0; k < f->queue.size(); ++k) {
        stamps[k] = f->queue[k].stamp;
        for (int i = 0; i < 3; ++i) {
            w[3 * k + i] = f->queue[k].w[i];
            a[3 * k + i] = f->queue[k].a[i];
        }
    }
    if (withVision) stamps[K - 1] = visionStamp;
    f->queue.clear();
    int status[kBurstMax] = {};
    {
        CurStream cs(f, f->sMain);
        Phase ph(f, 0);
        RC(eqf_tiled_propagate_burst(f->t, K, stamps, w, a, withVision ? 1 : 0, f->haveGeo ? f->Sll.p : nullptr, f->haveGeo ? f->Sll.ld : 0, status));
    }
    if (withVision && status[K - 1] == 0) {
        f->mirrorTime = visionStamp;
        f->mirrorValid = true;
    }
    if (visionStatus) *visionStatus = status[K - 1];
    return EQF_OK;
}

// removeOldLandmarks, removeOutliers, addNewLandmarks (VIOFilter.cpp:242-249, :345-443) on slots.  Fills the bearings in SLOT order (holes
// carry a dummy the device ignores); *haveUpdate = false when no landmark is left to update with.
int churn(eqf_tf* f, int n, const int* ids, const double* y, std::vector<double>* ySlots, bool* haveUpdate) {
    *haveUpdate = false;
    const std::vector<int>& have = f->ids;
    const int nh = (int)have.size();
    std::vector<int> posc(nh, 0);
    std::vector<char> seen(nh, 0), keep;
    for (int i = 0; i < nh; ++i) {  // ids ascending: where each state id sits in the measurement, if it does (removeOldLandmarks :393-419)
        const int pos = int(std::lower_bound(ids, ids + n, have[i]) - ids);
        posc[i] = std::min(pos, std::max(n - 1, 0));
        seen[i] = n > 0 && ids[posc[i]] == have[i];
    }
    std::unordered_set<int> haveSet(have.begin(), have.end());
    std::vector<int> newK;  // measurement entries without a landmark, ascending ids (:211-230 puts them last)
    for (int k = 0; k < n; ++k)
        if (!haveSet.count(ids[k])) newK.push_back(k);
    keep = seen;
    bool anySeen = false;
    for (char s : seen) anySeen = anySeen || s;
    double depth = f->set.initialSceneDepth;
    const double thr = f->set.outlierThreshold;
    const bool gate = thr < 2.0 && anySeen;  // (no chord of unit vectors is longer than 2: such a threshold switches the gate off, no readback)
    if (gate || (!newK.empty() && anySeen)) {
        const int Ns = eqf_tiled_num_landmarks(f->t);
        std::vector<double> p((size_t)3 * std::max(Ns, 1));
        RC(eqf_tiled_get_state_estimate(f->t, nullptr, nullptr, nullptr, p.data()));
        if (gate)  // removeOutliers :429-443: chord between the measured and the expected bearing
            for (int i = 0; i < nh; ++i) {
                const double* pi = &p[(size_t)3 * f->slotOf[i]];
                const double nrm = std::sqrt(pi[0] * pi[0] + pi[1] * pi[1] + pi[2] * pi[2]);
                double c2 = 0.0;
                for (int c = 0; c < 3; ++c) {
                    const double d = y[(size_t)3 * posc[i] + c] - pi[c] / nrm;
                    c2 += d * d;
                }
                if (std::sqrt(c2) > thr) keep[i] = 0;
            }
        bool anyKeep = false;
        for (char kq : keep) anyKeep = anyKeep || kq;
        if (!newK.empty() && anyKeep) {  // median scene depth of what is left, :353-366 (nth_element at size / 2)
            std::vector<double> d2;
            for (int i = 0; i < nh; ++i)
                if (keep[i]) {
                    const double* pi = &p[(size_t)3 * f->slotOf[i]];
                    d2.push_back(pi[0] * pi[0] + pi[1] * pi[1] + pi[2] * pi[2]);
                }
            std::nth_element(d2.begin(), d2.begin() + d2.size() / 2, d2.end());
            depth = std::sqrt(d2[d2.size() / 2]);
        }
    }
    int nOld = 0, nOut = 0;
    std::vector<int> removeSlots;
    for (int i = 0; i < nh; ++i) {
        if (!seen[i]) ++nOld;
        if (seen[i] && !keep[i]) ++nOut;
        if (!keep[i]) removeSlots.push_back(f->slotOf[i]);
    }
    std::vector<char> taken = f->taken;
    for (int s : removeSlots) taken[s] = 0;
    std::vector<int> addSlots;  // lowest free slots first: holes are refilled before the partition grows
    for (int s = 0; s < f->cap && addSlots.size() < newK.size(); ++s)
        if (!taken[s]) addSlots.push_back(s);
    if (addSlots.size() < newK.size()) {
        f->lastError = "more landmarks in view than the partitioned filter was created for";
        return EQF_ERR_CAPACITY;
    }
    for (int s : addSlots) taken[s] = 1;
    int top = -1;
    for (int s = 0; s < f->cap; ++s)
        if (taken[s]) top = s;
    const int nslots = std::max(top + 1, 1);
    if (!f->haveIds && newK.empty()) return EQF_OK;  // nothing yet, nothing to add: no storage either
    if (!removeSlots.empty() || !addSlots.empty()) {
        RC(setSlots(f, std::max(f->nslots, nslots)));
        std::vector<double> yNew((size_t)3 * std::max<size_t>(newK.size(), 1));
        for (size_t j = 0; j < newK.size(); ++j)
            for (int c = 0; c < 3; ++c) yNew[3 * j + c] = y[(size_t)3 * newK[j] + c];
        RC(eqf_tiled_edit_landmarks(f->t, (int)removeSlots.size(), removeSlots.data(), (int)addSlots.size(), addSlots.data(), yNew.data(), depth, nslots,
            f->Sll.p, f->Sll.ld));
        RC(setSlots(f, nslots));
        std::vector<int> nid, nsl;
        for (int i = 0; i < nh; ++i)
            if (keep[i]) {
                nid.push_back(have[i]);
                nsl.push_back(f->slotOf[i]);
            }
        std::vector<int> poscKeep;
        for (int i = 0; i < nh; ++i)
            if (keep[i]) poscKeep.push_back(posc[i]);
        for (size_t j = 0; j < newK.size(); ++j) {
            nid.push_back(ids[newK[j]]);
            nsl.push_back(addSlots[j]);
        }
        // (posc of the kept landmarks is needed below: keep it aligned with the new id list)
        posc = poscKeep;
        f->ids = nid;
        f->slotOf = nsl;
        f->haveIds = true;
        f->taken = taken;
        f->nslots = nslots;
        f->stats[0] += nOld;
        f->stats[1] += nOut;
        f->stats[2] += (long long)newK.size();
    } else {
        std::vector<int> poscKeep;
        for (int i = 0; i < nh; ++i)
            if (keep[i]) poscKeep.push_back(posc[i]);
        posc = poscKeep;
    }
    if (f->ids.empty()) return EQF_OK;
    // the measurement in slot order: landmarks that stayed, then the new ones (:211-230 matchMeasurementsToState)
    ySlots->assign((size_t)3 * f->nslots, 0.0);
    for (int s = 0; s < f->nslots; ++s) (*ySlots)[3 * s + 2] = 1.0;
    const size_t nKept = posc.size();
    for (size_t i = 0; i < f->ids.size(); ++i) {
        const int src = i < nKept ? posc[i] : newK[i - nKept];
        for (int c = 0; c < 3; ++c) (*ySlots)[(size_t)3 * f->slotOf[i] + c] = y[(size_t)3 * src + c];
    }
    *haveUpdate = true;
    return EQF_OK;
}

int checkPivots(eqf_tf* f, int* bad) {
    f->framesSinceCheck = 0;
    int v = 0;
    HIPC(hipStreamSynchronize(f->sMain));
    if (f->sAux) HIPC(hipStreamSynchronize(f->sAux));
    if (f->sSide) HIPC(hipStreamSynchronize(f->sSide));
    if (f->sAuxSide) HIPC(hipStreamSynchronize(f->sAuxSide));
    for (hipStream_t ps : {f->sPanel[0], f->sPanel[1], f->sComm})
        if (ps) HIPC(hipStreamSynchronize(ps));
    HIPC(hipMemcpy(&v, f->info, sizeof(int), hipMemcpyDeviceToHost));
    if (v) HIPC(hipMemset(f->info, 0, sizeof(int)));
    if (bad) *bad = v;
    return EQF_OK;
}

// ---- the update (VIOFilter.cpp:264-297): everything that is ENQUEUED, on the streams of the handle (current stream: main).  The frame's
// bearings are in the per-rank handle's pinned staging buffer already (eqf_tiled_stage_bearings).
int enqueueUpdate(eqf_tf* f) {
    const Geo& geo = f->geo;
    f->evNext = 0;
    {
        Phase ph(f, 2);
        // E and M are formed from the PRE-update Sigma (VIOFilter.cpp:285 before :297)
        RC(eqf_tiled_update_prep(f->t, nullptr, f->Sll.p, f->Sll.ld, f->M.p, f->M.ld, f->E.p, f->E.ld, f->G11));
    }
    const int nA = 2 * geo.nlc;
    RC(zero2d(f, f->accS));
    RC(zero2d(f, f->accE));
    hipEvent_t prepared = record(f);
    Chain S{f, f->M, 2, nA, 0, &f->wmaxS, &f->bufS, f->sSide, f->sPanel[0], prepared};
    Chain E{f, f->E, 3, 3 * geo.nlc, 1, &f->wmaxE, &f->bufE, f->sAuxSide, f->sPanel[1], prepared};
    S.hook = [f](int k, int bk, View Bop, int off, const Contributions& con) -> int {
        // Bop[:, off:] = [Y_k (3 nlc) | Yn_k (18)] of my process column; the rank's share of the downdate and of the reductions
        const Geo& g = f->geo;
        const View Yw = Bop.cols(off, off + 3 * g.nlc), Yn = Bop.cols(off + 3 * g.nlc, off + 3 * g.nlc + kNarrowS);
        const int r0 = 2 * k * g.bl;
        RC(copy2d(f, f->Yc.rows(r0, r0 + bk), Yw));
        if (!f->symmetric) {
            View YI;
            RC(rowsOperand(f, con, 3, [&g](int c, int wc) { return std::make_pair(wc - 3 * g.ncolsOf(c) - kNarrowS, 0); }, bk, f->aopW.p, true, 0, &YI));
            if (!YI.empty()) RC(copy2d(f, f->Yr.rows(r0, r0 + bk), YI));
        }
        if (k < (int)f->yReady.size()) f->yReady[k] = record(f);  // (Y_k is in place: its share of the downdate may start)
        return gemmTn(f, f->accS, Yn, Bop.cols(off, Bop.c), 1.0);  // [Sigma_b's downdate ; gamma_L ; .. | Gnn] += Yn_k^T [Y_k | Yn_k]
    };
    E.hook = [f](int, int, View Bop, int off, const Contributions&) -> int {
        const View En = Bop.cols(off, off + kNarrowE);
        return gemmTn(f, f->accE, En, En, 1.0);
    };
    // the E-chain (bundleLift's weights) needs nothing of the S-chain: it runs on its own stream pair next to it.  It is bound by its serial
    // diagonal blocks, the S-chain and the downdate by the matrix cores -- side by side they take little more than the longer one.  The two
    // are enqueued ALTERNATELY, block row by block row, so that neither stream waits for the host to be through with the other chain.
    hipEvent_t eDone = nullptr;
    const bool overlap = f->overlapChains != 0;
    int ddRows = 0;  // rows of Y whose share of the downdate has been issued already (on the E-chain's stream)
    if (overlap) {
        Phase phS(f, 3);
        Phase* phE = nullptr;
        {
            CurStream cs(f, f->sAux);
            wait(f, prepared);
            phE = new Phase(f, 4);
        }
        int rc = EQF_OK;
        // block rows whose share of the downdate goes to the E-chain's stream (see ddEarly)
        const int nEarly = (f->ddSlices == 0 && geo.nlr && geo.nlc) ? std::min(geo.nb, (geo.nb * std::max(0, std::min(f->ddEarly, 100)) + 50) / 100) : 0;
        auto earlyShare = [&](int kb) -> int {  // Sigma -= Y_kb^T Y_kb on the E-chain's stream, behind the hook of S's block row kb
            const int r0 = 2 * kb * geo.bl, r1 = r0 + 2 * geo.blockSize(kb);
            CurStream cs(f, f->sAux);
            wait(f, f->yReady[kb]);
            if (f->symmetric) {
                const int w3 = 3 * geo.bl;
                const int mask[8] = {w3, w3, 0, 1, 0, 0, 1, 0};
                return gemmTn(f, f->Sll, f->Yr.rows(r0, r1), f->Yc.rows(r0, r1), -1.0, mask);
            }
            return gemmTn(f, f->Sll, f->Yr.rows(r0, r1), f->Yc.rows(r0, r1), -1.0);
        };
        f->yReady.assign(geo.nb, nullptr);
        while (!rc && !(E.done() && S.done())) {
            if (!E.done()) {
                CurStream cs(f, f->sAux);
                rc = E.step();
            }
            if (!rc && !S.done()) {
                rc = S.step();
                // (one block row behind: the E-chain's stream should find Y ready, not wait for it)
                if (!rc && S.k >= 2 && S.k - 2 < nEarly) {
                    rc = earlyShare(S.k - 2);
                    ddRows = 2 * geo.bl * (S.k - 1);
                }
            }
        }
        if (!rc && nEarly >= geo.nb && geo.nb >= 1) {  // (everything early: the last block row's share too)
            rc = earlyShare(geo.nb - 1);
            ddRows = f->Yr.r;
        }
        {
            CurStream cs(f, f->sAux);
            phE->exit();
            delete phE;
            eDone = record(f);
        }
        phS.exit();
        if (rc) return rc;
    } else {
        Phase ph(f, 3);
        while (!S.done()) RC(S.step());
    }
    {
        Phase ph(f, 5);
        // Sigma_IJ -= Y_I^T Y_J (VIOFilter.cpp:297), one product; on a symmetric rank only the blocks on and above the block diagonal are
        // computed and the rest is mirrored
        if (geo.nlr && geo.nlc) {
            const int w3 = 3 * geo.bl;
            if (f->ddSlices > 0) {
                // (the integer pipe: same product, Y's columns cut into slices; on a symmetric rank Yr IS Yc -- one split)
                const bool same = f->Yr.p == f->Yc.p && f->Yr.c == f->Yc.c && f->Yr.ld == f->Yc.ld;
                const size_t need = eqf_tile_i8_workspace_bytes(f->Sll.r, f->Sll.c, f->Yr.r, f->ddSlices, same ? 1 : 0);
                if (need > f->i8WorkBytes) {
                    HIPC(hipStreamSynchronize(f->cur));
                    if (f->i8Work) (void)hipFree(f->i8Work);
                    f->i8Work = nullptr;
                    f->i8WorkBytes = 0;
                    HIPC(hipMalloc(&f->i8Work, need));
                    f->i8WorkBytes = need;
                }
                RC(eqf_tile_downdate_i8(f->device, f->cur, f->Sll.p, f->Sll.ld, f->Sll.r, f->Sll.c, f->Yr.p, f->Yr.ld, f->Yc.p, f->Yc.ld, f->Yr.r,
                    f->ddSlices, f->symmetric ? w3 : 0, f->i8Work, f->i8WorkBytes));
                if (f->symmetric) RC(eqf_tile_mirror(f->device, f->cur, f->Sll.p, f->Sll.ld, f->Sll.r, w3));
            } else if (f->symmetric) {
                const int mask[8] = {w3, w3, 0, 1, 0, 0, 1, 0};
                if (ddRows < f->Yr.r) RC(gemmTn(f, f->Sll, f->Yr.rows(ddRows, f->Yr.r), f->Yc.rows(ddRows, f->Yc.r), -1.0, mask));
                if (ddRows > 0) wait(f, eDone);  // (the shares on the E-chain's stream: the mirror reads what they wrote)
                RC(eqf_tile_mirror(f->device, f->cur, f->Sll.p, f->Sll.ld, f->Sll.r, w3));
            } else {
                if (ddRows < f->Yr.r) RC(gemmTn(f, f->Sll, f->Yr.rows(ddRows, f->Yr.r), f->Yc.rows(ddRows, f->Yc.r), -1.0));
            }
        }
    }
    if (overlap) {
        wait(f, eDone);
    } else {
        Phase ph(f, 4);
        while (!E.done()) RC(E.step());
    }
    {
        Phase ph(f, 6);
        View acc;
        RC(gatherColumns(f, f->accS.cols(0, 3 * geo.nlc), &acc));
        RC(copy2d(f, flat(f->GnnBuf, kNarrowS, kNarrowS), f->accS.cols(3 * geo.nlc, 3 * geo.nlc + kNarrowS)));
        hipLaunchKernelGGL(k_tf_add, dim3(1), dim3(128), 0, f->cur, f->G11sum, f->G11, f->accE.p, 121);
        RC(eqf_tiled_update_finish(f->t, acc.p, acc.ld, f->GnnBuf, f->G11sum));
    }
    HIPC(hipGetLastError());
    return EQF_OK;
}

int update(eqf_tf* f, const std::vector<double>& y) {
    RC(eqf_tiled_stage_bearings(f->t, y.data()));
    const bool graphable = f->graphs && f->world == 1 && !f->profiling;
    const std::pair<int, int> key(f->geo.N, eqf_tiled_pingpong(f->t));
    hipGraphExec_t exec = nullptr;
    if (graphable) {
        auto it = f->graphExec.find(key);
        if (it != f->graphExec.end()) {
            exec = it->second;
        } else if (f->graphSeen[key]++ >= 1) {
            // capture: the forks to the other three streams hang off events recorded on main, and every one of them is joined again before
            // the update's last launches (the E-chain's end event, the look-ahead events the next block row waits for)
            hipGraph_t graph = nullptr;
            HIPC(hipStreamBeginCapture(f->sMain, hipStreamCaptureModeThreadLocal));
            const int rc = enqueueUpdate(f);
            const hipError_t e = hipStreamEndCapture(f->sMain, &graph);
            if (rc || e != hipSuccess || !graph) {
                if (graph) hipGraphDestroy(graph);
                f->graphs = 0;  // (not capturable here: plain launches from now on)
                if (rc) return rc;
            } else {
                if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
                hipGraphDestroy(graph);
                if (exec) f->graphExec[key] = exec;
                else f->graphs = 0;
            }
        }
    }
    if (exec) {
        HIPC(hipGraphLaunch(exec, f->sMain));
        RC(eqf_tiled_bearings_consumed(f->t, f->sMain));  // (the graph's copy node of the staged bearings: see eqf_tiled.hip)
        ++f->graphLaunches;
    } else {
        RC(enqueueUpdate(f));
    }
    if (f->checkEvery && ++f->framesSinceCheck >= f->checkEvery) {
        int bad = 0;
        RC(checkPivots(f, &bad));
        if (bad) {
            f->lastError = "a pivot of S or Sigma_e was not positive (distributed factorisation)";
            return EQF_ERR_NUMERIC;
        }
    }
    return EQF_OK;
}

// Dense Sigma over ALL slots in use (slot order, holes included), gathered to every rank: S is (11 + 3 nslots)^2, host, row-major
int slotCovariance(eqf_tf* f, std::vector<double>* S, int* nOut) {
    const Geo& geo = f->geo;
    const int N = geo.N, n = 11 + 3 * N;
    *nOut = n;
    S->assign((size_t)n * n, 0.0);
    std::vector<double> base((size_t)11 * n);
    RC(eqf_tiled_get_base(f->t, base.data(), n));
    for (int r = 0; r < 11; ++r)
        for (int c = 0; c < n; ++c) {
            (*S)[(size_t)c * n + r] = base[(size_t)r * n + c];  // (only the base ROWS are kept: the columns are their transpose)
        }
    for (int r = 0; r < 11; ++r)
        for (int c = 0; c < n; ++c) (*S)[(size_t)r * n + c] = base[(size_t)r * n + c];
    int rmax = 1, cmax = 1;
    for (int r = 0; r < f->Pr; ++r) rmax = std::max(rmax, 3 * geo.nrowsOf(r));
    for (int c = 0; c < f->Pc; ++c) cmax = std::max(cmax, 3 * geo.ncolsOf(c));
    double *mine = nullptr, *other = nullptr;
    HIPC(hipMalloc(reinterpret_cast<void**>(&mine), sizeof(double) * rmax * cmax));
    if (f->world > 1) HIPC(hipMalloc(reinterpret_cast<void**>(&other), sizeof(double) * rmax * cmax));
    int rc = EQF_OK;
    {
        CurStream cs(f, f->sMain);
        const View pad = flat(mine, rmax, cmax);
        rc = zero2d(f, pad);
        if (!rc && geo.nlr && geo.nlc) rc = copy2d(f, pad.sub(0, 3 * geo.nlr, 0, 3 * geo.nlc), f->Sll);
        std::vector<double> h((size_t)rmax * cmax);
        for (int rank = 0; rank < f->world && !rc; ++rank) {
            double* buf = rank == f->rank ? mine : other;
            rc = bcast(f, 2, 0, rank, buf, (size_t)rmax * cmax);
            if (rc) break;
            if (hipStreamSynchronize(f->cur) != hipSuccess || hipMemcpy(h.data(), buf, sizeof(double) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) {
                rc = EQF_ERR_HIP;
                break;
            }
            const int r = rank / f->Pc, c = rank % f->Pc;
            const Geo og(N, f->bl, f->Pr, f->Pc, r, c);
            for (int i = 0; i < 3 * og.nlr; ++i) {
                const size_t gi = 11 + 3 * (size_t)og.rowMap[i / 3] + i % 3;
                for (int j = 0; j < 3 * og.nlc; ++j) (*S)[gi * n + 11 + 3 * (size_t)og.colMap[j / 3] + j % 3] = h[(size_t)i * cmax + j];
            }
        }
    }
    hipFree(mine);
    if (other) hipFree(other);
    return rc;
}

// The panel streams (per chain: the next block row's solve and exchanges) share the main streams' CU set; the exchange stream carries no
// kernels of ours.  Made when "panel_ahead" is on -- from two ranks on, or by option: a CU-masked stream is a hardware queue of its own,
// and the GPU slows down once a process has used more than a handful of them (scripts/handle_age_probe.py, eqf_stream_create_masked), so
// the one-rank filter holds four streams, not seven.
int panelStreams(eqf_tf* f) {
    for (int i = 0; i < 3; ++i) {
        hipStream_t* dst = i < 2 ? &f->sPanel[i] : &f->sComm;
        if (*dst) continue;
        void* ps = nullptr;
        // plain streams unless EQF_TILED_PANEL_MASKED=1: with three more CU-masked queues the process is past what the GPU serves at full speed
        // (one rank, "panel_ahead" = 1, N = 4000: 85 - 99 ms a frame masked, 80 - 82 plain: profiles/r06_handle_age.txt)
        const char* e = std::getenv("EQF_TILED_PANEL_MASKED");
        const bool masked = e && std::atoi(e) != 0;
        if (f->reserve > 0 && masked) RC(eqf_stream_create_masked(f->device, 0, f->reserve, 1, &ps));
        else {
            hipStream_t st;
            HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            ps = st;
        }
        *dst = (hipStream_t)ps;
    }
    return EQF_OK;
}

void freeAll(eqf_tf* f) {
    if (!f) return;
    hipSetDevice(f->device);
    for (hipStream_t s : {f->sMain, f->sSide, f->sAux, f->sAuxSide, f->sPanel[0], f->sPanel[1], f->sComm})
        if (s) hipStreamSynchronize(s);
    if (f->t) eqf_tiled_destroy(f->t);
    for (double* p : f->allocs) hipFree(p);
    if (f->info) hipFree(f->info);
    if (f->i8Work) hipFree(f->i8Work);
    for (void* w : f->chainWork)
        if (w) hipFree(w);
    for (auto& g : f->graphExec) hipGraphExecDestroy(g.second);
    for (hipEvent_t e : f->evPool) hipEventDestroy(e);
    for (auto& p : f->phasePending) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    if (f->ownStreams)
        for (hipStream_t s : {f->sMain, f->sSide, f->sAux, f->sAuxSide, f->sPanel[0], f->sPanel[1], f->sComm})
            if (s) eqf_stream_destroy(f->device, s);
    delete f;
}
}  // namespace

extern "C" {

int eqf_tf_create(const eqf_settings* settings, int capacity_landmarks, int block_landmarks, int Pr, int Pc, int rank, int device, int reserve_cus,
    const eqf_tf_comm* comm, eqf_tf** out) {
    if (!settings || !out || capacity_landmarks < 1 || block_landmarks < 1 || Pr < 1 || Pc < 1 || Pc % Pr != 0 || rank < 0 || rank >= Pr * Pc) return EQF_ERR_INVALID;
    if (Pr * Pc > 1 && (!comm || !comm->bcast)) return EQF_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return EQF_ERR_NO_DEVICE;
    DeviceScope ds(device);
    if (!ds.ok) return EQF_ERR_HIP;
    eqf_tf* f = new eqf_tf();
    f->cap = capacity_landmarks;
    f->bl = block_landmarks;
    f->Pr = Pr;
    f->Pc = Pc;
    f->rank = rank;
    f->world = Pr * Pc;
    f->pr = rank / Pc;
    f->pc = rank % Pc;
    f->device = device;
    f->set = *settings;
    if (comm) {
        f->comm = *comm;
        f->haveComm = true;
    }
    f->taken.assign(f->cap, 0);
    if (const char* e = std::getenv("EQF_TILED_RESERVE_CUS")) reserve_cus = reserve_cus < 0 ? std::atoi(e) : reserve_cus;
    // (round 6, N = 4000 on one rank, profiles/r06_tiled_reserve_sweep.txt: 0 / 4 / 8 / 12 / 16 / 24 / 32 / 48 reserved CUs -> 157.7 / 153.3 / 162.7 /
    // 166.3 / 169.2 / 169.5 / 169.3 / 150.7 steps/s, the monolithic single-GPU path 169: with 8 the E-chain's look-ahead factorisations were its
    // critical path; from 16 on the partitioned filter is level with the monolithic one)
    f->reserve = reserve_cus < 0 ? 24 : reserve_cus;
    // On ONE rank the two chains of an update run side by side (measured: profiles/r05_tiled_*).  Over a collective library whose kernels
    // share the GPU with ours (RCCL) they run one after the other unless the caller asks for the overlap (option "overlap_chains" /
    // EQF_TILED_OVERLAP_CHAINS=1): every exchange already goes through one stream in program order, but the interleaved form has only ever
    // run over gloo with the ranks on one GPU, never on a node -- unvalidated, so not the default there.
    f->overlapChains = f->world > 1 ? 0 : 1;
    if (const char* e = std::getenv("EQF_TILED_OVERLAP_CHAINS")) f->overlapChains = std::atoi(e) != 0;
    // Block row k + 1 solved and exchanged NEXT TO the products of block row k (Chain::step): what hides the broadcasts on a node.  On one
    // rank there is nothing to hide and the extra streams cost: measured on the MI355X, 1 x 1 grid (profiles/r05_tiled_host_loop.txt), N = 4000
    // 68.0 -> 76.3 ms per frame, N = 1000 5.3 -> 9.4 ms (two more streams, three cross-stream events and two extra product launches per
    // block row and chain) -- so it is on from two ranks on, and an option ("panel_ahead") everywhere.
    f->panelAhead = f->world > 1 ? 1 : 0;
    int rc = eqf_tiled_create(settings, f->cap, device, &f->t);
    if (!rc && hipMalloc(reinterpret_cast<void**>(&f->info), sizeof(int)) != hipSuccess) rc = EQF_ERR_HIP;
    if (!rc && hipMemset(f->info, 0, sizeof(int)) != hipSuccess) rc = EQF_ERR_HIP;
    // Two stream pairs with disjoint CU sets: `reserve` CUs for the look-ahead factorisation of the next diagonal block (side streams), all the
    // others for everything else (main streams).  reserve = 0: plain streams -- the look-ahead only runs when the trailing update happens
    // to leave room.
    if (!rc) {
        void* s[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < 4 && !rc; ++i) {
            if (f->reserve > 0) rc = eqf_stream_create_masked(device, 0, f->reserve, i % 2 == 0 ? 1 : 0, &s[i]);
            else {
                hipStream_t st;
                if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) rc = EQF_ERR_HIP;
                s[i] = st;
            }
        }
        f->sMain = (hipStream_t)s[0];
        f->sSide = (hipStream_t)s[1];
        f->sAux = (hipStream_t)s[2];
        f->sAuxSide = (hipStream_t)s[3];
        f->ownStreams = true;
        f->cur = f->sMain;
        // (the panel streams and the exchange stream: only where there is an exchange to hide -- panelStreams)
        if (!rc && f->panelAhead) rc = panelStreams(f);
    }
    if (!rc) rc = eqf_tiled_set_stream(f->t, f->sMain);
    if (rc) {
        freeAll(f);
        return rc;
    }
    *out = f;
    return EQF_OK;
}

void eqf_tf_destroy(eqf_tf* f) {
    if (f) freeAll(f);
}

int eqf_tf_set_option(eqf_tf* f, const char* name, int value) {
    if (!f || !name) return EQF_ERR_INVALID;
    const std::string n(name);
    if (n == "lookahead") f->lookahead = value;
    else if (n == "overlap_chains") f->overlapChains = value;
    else if (n == "downdate_slices") {
        if (value != 0 && (value < 5 || value > 7)) return EQF_ERR_INVALID;
        f->ddSlices = value;
    }
    else if (n == "solve_inverse") f->solveInverse = value ? 1 : 0;
    else if (n == "downdate_early") {
        if (value < 0 || value > 100) return EQF_ERR_INVALID;
        f->ddEarly = value;
    }
    else if (n == "chain_slices") {
        if (value != 0 && (value < 5 || value > 7)) return EQF_ERR_INVALID;
        f->chainSlices = value;
    }
    else if (n == "trsm_leaf") {
        if (value < 0 || value > 16) return EQF_ERR_INVALID;
        f->trsmLeaf = value;
    }
    else if (n == "burst") f->burst = value;
    else if (n == "check_every") f->checkEvery = value;
    else if (n == "profiling") f->profiling = value;
    else if (n == "graphs") f->graphs = value;
    else if (n == "panel_ahead") {
        if (value && f->ownStreams) {
            DeviceScope ds(f->device);
            RC(panelStreams(f));
        }
        f->panelAhead = value;
    }
    else return EQF_ERR_INVALID;
    return EQF_OK;
}

// VIOFilter::processIMUData (VIOFilter.cpp:120-131).  IMU calls are QUEUED (up to 15 of them) and leave for the device with the next vision
// call, getter or full queue as one burst: every call keeps its own linearisation, but the local blocks of Sigma are read and written once
// per burst (eqf_tiled_propagate_burst).  The status returned is the reference's control flow (:120-131, :146-152) on the stamps.
int eqf_tf_process_imu(eqf_tf* f, double stamp, const double* omega, const double* accel) {
    if (!f || !omega || !accel) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    if (!ds.ok) return EQF_ERR_HIP;
    if (!f->burst) {
        CurStream cs(f, f->sMain);
        Phase ph(f, 0);
        return eqf_tiled_propagate(f->t, stamp, omega, accel, 1, f->haveGeo ? f->Sll.p : nullptr, f->haveGeo ? f->Sll.ld : 0);
    }
    if (!f->mirrorValid) {
        RC(eqf_tiled_get_time(f->t, &f->mirrorTime));
        f->mirrorValid = true;
    }
    const int st = f->mirrorTime < 0 ? EQF_SKIPPED_BEFORE_FIRST_IMU : (!(stamp - f->mirrorTime > 0) ? EQF_SKIPPED_NONPOSITIVE_DT : EQF_OK);
    f->mirrorTime = stamp;
    eqf_tf::Rec r;
    r.stamp = stamp;
    for (int i = 0; i < 3; ++i) {
        r.w[i] = omega[i];
        r.a[i] = accel[i];
    }
    f->queue.push_back(r);
    if ((int)f->queue.size() >= kBurstMax - 1) RC(flushQueue(f, false, 0.0, nullptr));
    return st;
}

// VIOFilter::processVisionData (VIOFilter.cpp:232-302): ids strictly ascending (:239-240), bearings[n][3] unit vectors, host memory.
int eqf_tf_process_vision(eqf_tf* f, double stamp, int n, const int* ids, const double* bearings) {
    if (!f || n < 0 || (n > 0 && (!ids || !bearings))) return EQF_ERR_INVALID;
    for (int k = 1; k < n; ++k)
        if (ids[k] <= ids[k - 1]) return EQF_ERR_UNSORTED;
    DeviceScope ds(f->device);
    if (!ds.ok) return EQF_ERR_HIP;
    CurStream cs(f, f->sMain);
    int st = 0;
    if (f->burst) {
        RC(flushQueue(f, true, stamp, &st));  // the queued IMU calls + :233 integrateUpToTime, one pass over the local blocks
    } else {
        Phase ph(f, 0);
        st = eqf_tiled_propagate(f->t, stamp, nullptr, nullptr, 0, f->haveGeo ? f->Sll.p : nullptr, f->haveGeo ? f->Sll.ld : 0);
        if (st < 0) return st;
    }
    if (st != 0) return st;  // :234-236
    std::vector<double> ySlots;
    bool have = false;
    {
        Phase ph(f, 1);
        RC(churn(f, n, ids, bearings, &ySlots, &have));  // :242-249
    }
    if (!have) return EQF_SKIPPED_NO_BEARINGS;  // :258-259
    return update(f, ySlots);
}

int eqf_tf_synchronize(eqf_tf* f) {
    if (!f) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    RC(flushQueue(f, false, 0.0, nullptr));
    for (hipStream_t s : {f->sMain, f->sSide, f->sAux, f->sAuxSide, f->sPanel[0], f->sPanel[1], f->sComm})
        if (s) HIPC(hipStreamSynchronize(s));
    return EQF_OK;
}

int eqf_tf_check(eqf_tf* f) {
    if (!f) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    int bad = 0;
    RC(checkPivots(f, &bad));
    return bad ? EQF_ERR_NUMERIC : EQF_OK;
}

int eqf_tf_device_error(eqf_tf* f) {
    if (!f) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    RC(flushQueue(f, false, 0.0, nullptr));
    return eqf_tiled_device_error(f->t);
}

int eqf_tf_num_landmarks(eqf_tf* f) { return f ? (int)f->ids.size() : EQF_ERR_INVALID; }
int eqf_tf_num_slots(eqf_tf* f) { return f ? f->nslots : EQF_ERR_INVALID; }

int eqf_tf_get_ids(eqf_tf* f, int* ids, int* slots) {
    if (!f) return EQF_ERR_INVALID;
    if (ids) std::copy(f->ids.begin(), f->ids.end(), ids);
    if (slots) std::copy(f->slotOf.begin(), f->slotOf.end(), slots);
    return EQF_OK;
}

int eqf_tf_get_time(eqf_tf* f, double* time) {
    if (!f || !time) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    RC(flushQueue(f, false, 0.0, nullptr));
    return eqf_tiled_get_time(f->t, time);
}

// VIOFilter::stateEstimate (:304): landmarks in the reference's order
int eqf_tf_get_state_estimate(eqf_tf* f, double* pose_q, double* pose_x, double* velocity, double* p) {
    if (!f) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    RC(flushQueue(f, false, 0.0, nullptr));
    const int Ns = eqf_tiled_num_landmarks(f->t);
    std::vector<double> ps((size_t)3 * std::max(Ns, 1));
    RC(eqf_tiled_get_state_estimate(f->t, pose_q, pose_x, velocity, ps.data()));
    if (p)
        for (size_t i = 0; i < f->ids.size(); ++i)
            for (int c = 0; c < 3; ++c) p[3 * i + c] = ps[(size_t)3 * f->slotOf[i] + c];
    return EQF_OK;
}

int eqf_tf_get_bias(eqf_tf* f, double* bias6) {
    if (!f || !bias6) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    RC(flushQueue(f, false, 0.0, nullptr));
    return eqf_tiled_get_bias(f->t, bias6);
}

// delta (2 N), gamma (11 + 3 N), Gamma (9 + 3 N) of the last update, landmarks in the reference's order
int eqf_tf_get_last_update(eqf_tf* f, double* delta, double* gamma, double* Gamma) {
    if (!f) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    const int Ns = eqf_tiled_num_landmarks(f->t);
    std::vector<double> d((size_t)2 * std::max(Ns, 1)), g((size_t)11 + 3 * Ns), G((size_t)9 + 3 * Ns);
    RC(eqf_tiled_get_last_update(f->t, d.data(), g.data(), G.data()));
    const size_t N = f->ids.size();
    if (gamma) std::copy(g.begin(), g.begin() + 11, gamma);
    if (Gamma) std::copy(G.begin(), G.begin() + 9, Gamma);
    for (size_t i = 0; i < N; ++i) {
        const size_t s = f->slotOf[i];
        if (delta)
            for (int c = 0; c < 2; ++c) delta[2 * i + c] = d[2 * s + c];
        if (gamma)
            for (int c = 0; c < 3; ++c) gamma[11 + 3 * i + c] = g[11 + 3 * s + c];
        if (Gamma)
            for (int c = 0; c < 3; ++c) Gamma[9 + 3 * i + c] = G[9 + 3 * s + c];
    }
    return EQF_OK;
}

// VIOFilter::stateCovariance (:306-309): dense Sigma, reference index map and landmark order (slot_order = 0), or over ALL slots in use,
// holes included (slot_order = 1; n = 11 + 3 * eqf_tf_num_slots).  Collective: every rank of the grid calls it.
int eqf_tf_get_sigma(eqf_tf* f, double* dst, int ld, int slot_order) {
    if (!f || !dst) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    RC(flushQueue(f, false, 0.0, nullptr));
    if (!f->haveGeo) {  // no landmarks yet: the base block
        if (ld < 11) return EQF_ERR_INVALID;
        std::vector<double> base((size_t)11 * 11);
        RC(eqf_tiled_get_base(f->t, base.data(), 11));
        for (int r = 0; r < 11; ++r) std::copy(base.begin() + r * 11, base.begin() + (r + 1) * 11, dst + (size_t)r * ld);
        return EQF_OK;
    }
    std::vector<double> S;
    int n = 0;
    RC(slotCovariance(f, &S, &n));
    if (slot_order) {
        if (ld < n) return EQF_ERR_INVALID;
        for (int r = 0; r < n; ++r) std::copy(S.begin() + (size_t)r * n, S.begin() + (size_t)(r + 1) * n, dst + (size_t)r * ld);
        return EQF_OK;
    }
    const int N = (int)f->ids.size(), m = 11 + 3 * N;
    if (ld < m) return EQF_ERR_INVALID;
    std::vector<int> idx(m);
    for (int i = 0; i < 11; ++i) idx[i] = i;
    for (int i = 0; i < N; ++i)
        for (int c = 0; c < 3; ++c) idx[11 + 3 * i + c] = 11 + 3 * f->slotOf[i] + c;
    for (int r = 0; r < m; ++r) {
        const double* src = &S[(size_t)idx[r] * n];
        double* d = dst + (size_t)r * ld;
        for (int c = 0; c < m; ++c) d[c] = src[idx[c]];
    }
    return EQF_OK;
}

// Restart from a snapshot (everything eqf_get_* / eqf_tf_get_* return can be fed back): N landmarks with ids (reference order), origin state,
// group element, bias, the DENSE Sigma (n x n, n = 11 + 3 N, every rank holds it once, here), the integrator's scalars.
int eqf_tf_set_state(eqf_tf* f, int N, const int* ids, const double* pose_q, const double* pose_x, const double* velocity, const double* p0,
    const double* A_q, const double* A_x, const double* w, const double* Q_q, const double* Q_a, const double* bias6, const double* sigma, int ld,
    double currentTime, const double* currentVelocity6, const double* accumulatedVelocity6, double accumulatedTime, int initialised) {
    if (!f || N < 0 || !sigma || ld < 11 + 3 * N || (N > 0 && !ids)) return EQF_ERR_INVALID;
    if (N > f->cap) return EQF_ERR_CAPACITY;
    DeviceScope ds(f->device);
    CurStream cs(f, f->sMain);
    f->queue.clear();
    f->mirrorValid = false;
    RC(setSlots(f, std::max(N, 1)));
    const Geo& geo = f->geo;
    if (N && geo.nlr && geo.nlc) {
        std::vector<double> h((size_t)3 * geo.nlr * 3 * geo.nlc);
        for (int i = 0; i < 3 * geo.nlr; ++i) {
            const size_t gi = 11 + 3 * (size_t)geo.rowMap[i / 3] + i % 3;
            for (int j = 0; j < 3 * geo.nlc; ++j) h[(size_t)i * 3 * geo.nlc + j] = sigma[gi * ld + 11 + 3 * (size_t)geo.colMap[j / 3] + j % 3];
        }
        HIPC(hipStreamSynchronize(f->cur));
        HIPC(hipMemcpy2D(f->Sll.p, sizeof(double) * f->Sll.ld, h.data(), sizeof(double) * 3 * geo.nlc, sizeof(double) * 3 * geo.nlc, 3 * geo.nlr,
            hipMemcpyHostToDevice));
    }
    RC(eqf_tiled_set_state(f->t, N, pose_q, pose_x, velocity, p0, A_q, A_x, w, Q_q, Q_a, bias6, sigma, ld, currentTime, currentVelocity6,
        accumulatedVelocity6, accumulatedTime, initialised));
    f->ids.assign(ids, ids + N);
    f->haveIds = true;
    f->slotOf.resize(N);
    for (int i = 0; i < N; ++i) f->slotOf[i] = i;  // the snapshot's order is the reference's: slot i = landmark i
    f->taken.assign(f->cap, 0);
    std::fill(f->taken.begin(), f->taken.begin() + N, 1);
    f->nslots = N;
    return EQF_OK;
}

// churn statistics: removed_old, removed_outliers, added (landmarks, since creation)
int eqf_tf_get_churn_stats(eqf_tf* f, long long* stats3) {
    if (!f || !stats3) return EQF_ERR_INVALID;
    std::copy(f->stats, f->stats + 3, stats3);
    return EQF_OK;
}

// the rank's local matrix (device memory owned by the handle): rows x cols view with leading dimension ld; NULL before any landmark exists
int eqf_tf_local_matrix(eqf_tf* f, double** ptr, int* rows, int* cols, int* ld) {
    if (!f) return EQF_ERR_INVALID;
    if (ptr) *ptr = f->haveGeo ? f->Sll.p : nullptr;
    if (rows) *rows = f->haveGeo ? f->Sll.r : 0;
    if (cols) *cols = f->haveGeo ? f->Sll.c : 0;
    if (ld) *ld = f->haveGeo ? f->Sll.ld : 0;
    return EQF_OK;
}

// GPU time per phase since profiling was switched on (eqf_tf_set_option "profiling"): propagate, churn, prep, chain_S, chain_E, downdate,
// finish (ms, 7 values; synchronises).  chain_E runs next to chain_S + downdate: the phases are stream-busy times, not a sum.
int eqf_tf_get_phases(eqf_tf* f, double* ms7) {
    if (!f || !ms7) return EQF_ERR_INVALID;
    DeviceScope ds(f->device);
    RC(eqf_tf_synchronize(f));
    for (auto& p : f->phasePending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) f->phaseMs[p.phase] += ms;
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    f->phasePending.clear();
    std::copy(f->phaseMs, f->phaseMs + kPhases, ms7);
    return EQF_OK;
}
const char* eqf_tf_phase_name(int i) { return (i >= 0 && i < kPhases) ? kPhaseNames[i] : "?"; }
const char* eqf_tf_last_error(eqf_tf* f) { return f ? f->lastError.c_str() : ""; }
void* eqf_tf_tiled_handle(eqf_tf* f) { return f ? f->t : nullptr; }
long long eqf_tf_graph_launches(eqf_tf* f) { return f ? f->graphLaunches : -1; }

}  // extern "C"
