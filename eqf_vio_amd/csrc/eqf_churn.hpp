// Landmark-set changes and read-out kernels (gfx950).
//
// Device side of VIOFilter::addNewLandmarks / removeLandmarkAtIndex / removeOutliers
// (eqf_vio/src/VIOFilter.cpp:345-443) and of VIOFilter::stateEstimate (:304).  The id matching itself is
// integer work on ids the host already holds (VIOFilter.cpp:211-230, :393-419) and stays on the host.
#pragma once
#include "eqf_device.hpp"
#include "eqf_math.hpp"
#include "eqf_propagate.hpp"
#include "eqf_update.hpp"
#include "eqf_handoff.hpp"

namespace eqf {

// Per-landmark probe of the current estimate: chord between the measured bearing and the predicted one
// (removeOutliers, VIOFilter.cpp:429-443) and squared depth (addNewLandmarks median, :357-366).
// Speculative outlier gate (gateFlag != nullptr): a landmark whose chord exceeds gateThr switches the filter's queued
// update off (updateOk = 0) and raises gateFlag[b] in pinned host memory -- the host, which enqueued the update without
// waiting for this answer, redoes the frame the slow way when it next touches the handle.
__global__ void k_probe(Glob* g, const double* p0, const double* Q, int cap, const double* bearings,
    long long bearStride, const int* perm, double* chord, double* depth2, double gateThr, int* gateFlag) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g[b].N) return;
    const double* P = p0 + (long long)b * 3 * cap;
    const double* q = Q + (long long)b * 5 * cap;
    const quat Qq = quat{q[i], q[cap + i], q[2 * cap + i], q[3 * cap + i]};
    const d3 qhat = scl(1.0 / q[4 * cap + i], qrot(qinv(Qq), mk3(P[i], P[cap + i], P[2 * cap + i])));
    depth2[(long long)b * cap + i] = dot3(qhat, qhat);
    double ch = 0.0;
    if (bearings) {
        const int k = perm ? perm[(long long)b * cap + i] : i;
        if (k >= 0) {
            const double* y = bearings + (long long)b * bearStride + 3 * k;
            ch = nrm3(sub(mk3(y[0], y[1], y[2]), unit3(qhat)));
        }
    }
    chord[(long long)b * cap + i] = ch;
    if (gateFlag && ch > gateThr) {
        g[b].updateOk = 0;
        gateFlag[b] = 1;
    }
}
// updateOk[b] = mask[b] (redo of a speculatively skipped update: only the flagged filters take part)
__global__ void k_set_update_ok(Glob* g, const int* mask, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) g[b].updateOk = mask[b] ? 1 : 0;
}

// Median scene depth for addNewLandmarks (VIOFilter.cpp:357-366): sqrt of the (N/2)-th order statistic of the squared
// depths k_probe wrote -- by rank counting, on the device, so that adding landmarks needs no readback.
__global__ void k_median_depth(const Glob* g, const double* depth2, int cap, double* sel) {
    const int b = blockIdx.y;
    const int N = g[b].N;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double* d = depth2 + (long long)b * cap;
    const double di = d[i];
    int rank = 0;
    for (int j = 0; j < N; ++j) rank += (d[j] < di) || (d[j] == di && j < i);
    if (rank == N / 2) sel[b] = sqrt(di);
}

// Remove landmarks: Sigma_out = Sigma_in with the rows/cols of dropped landmarks erased, map[b][newI] = oldI.
// Filters without removals pass the identity map (the ping-pong parity is shared by the whole batch).
template <typename T>
__device__ __forceinline__ void compactSigmaRow(const int* map, const int* newN, int cap, const T* Sin, T* Sout, long long sigmaStride, int ld) {
    const int b = blockIdx.z;
    const int Nn = newN[b];
    const int nvn = kLm0 + 3 * Nn;
    const int R = blockIdx.y;  // output row
    if (R >= nvn) return;
    const int* mp = map + (long long)b * cap;
    const int Rs = (R < kLm0) ? R : kLm0 + 3 * mp[(R - kLm0) / 3] + (R - kLm0) % 3;
    const T* src = Sin + (long long)b * sigmaStride + (long long)Rs * ld;
    T* dst = Sout + (long long)b * sigmaStride + (long long)R * ld;
    for (int Cc = blockIdx.x * blockDim.x + threadIdx.x; Cc < nvn; Cc += gridDim.x * blockDim.x) {
        const int Cs = (Cc < kLm0) ? Cc : kLm0 + 3 * mp[(Cc - kLm0) / 3] + (Cc - kLm0) % 3;
        dst[Cc] = src[Cs];
    }
}
constexpr int kLmRec = 3 + 5 + 15;  // scratch record per landmark: p0, Q, constants
// grid = (ceil(nvn / 256), nvn, B) over the output of the largest filter, block = 256
template <typename T>
__global__ __launch_bounds__(256) void k_compact(Glob* g, const int* map, const int* newN, int cap, const T* Sin, T* Sout, long long sigmaStride,
    int ld, double* p0, double* Q, double* lmc, double* scratch) {
    compactSigmaRow<T>(map, newN, cap, Sin, Sout, sigmaStride, ld);
    if (blockIdx.x != 0 || blockIdx.y != 0) return;
    const int b = blockIdx.z;
    const int* mp = map + (long long)b * cap;
    const int Nn = newN[b];
    // (all 23 loads of a landmark first, then its stores: with a store between two loads the compiler must assume they alias and the
    // record becomes 23 serial round trips -- this one workgroup was the long pole of the launch)
    for (int i = threadIdx.x; i < Nn; i += blockDim.x) {
        const int o = mp[i];
        double v[kLmRec];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = p0[((long long)b * 3 + c) * cap + o];
#pragma unroll
        for (int c = 0; c < 5; ++c) v[3 + c] = Q[((long long)b * 5 + c) * cap + o];
#pragma unroll
        for (int c = 0; c < 15; ++c) v[8 + c] = lmc[((long long)b * 15 + c) * cap + o];
#pragma unroll
        for (int c = 0; c < kLmRec; ++c) scratch[((long long)b * kLmRec + c) * cap + i] = v[c];
    }
    __threadfence_block();
    __syncthreads();
    for (int i = threadIdx.x; i < Nn; i += blockDim.x) {
        double v[kLmRec];
#pragma unroll
        for (int c = 0; c < kLmRec; ++c) v[c] = scratch[((long long)b * kLmRec + c) * cap + i];
#pragma unroll
        for (int c = 0; c < 3; ++c) p0[((long long)b * 3 + c) * cap + i] = v[c];
#pragma unroll
        for (int c = 0; c < 5; ++c) Q[((long long)b * 5 + c) * cap + i] = v[3 + c];
#pragma unroll
        for (int c = 0; c < 15; ++c) lmc[((long long)b * 15 + c) * cap + i] = v[8 + c];
    }
    if (threadIdx.x == 0) g[b].N = Nn;
}

// Append landmarks to filter b: p0 = bearing * depth, Q = identity, Sigma grows with zero cross terms and
// initialPointVariance on the new diagonal (VIOFilter.cpp:367-390).  The bearing of the j-th new landmark is measurement entry
// perm[b][nOld + j] (the permutation the update will use; nullptr: the identity).  depth: the median scene depth of the current estimate
// (:357-366) -- depthSel[b] if given (k_median_depth's result), else every workgroup selects it itself from the squared depths k_probe left
// (the same order statistic by the same rank counting; nOld <= kMedianInAppend) -- or initialSceneDepth for an empty filter.
constexpr int kMedianInAppend = 1024;
template <typename T>
__global__ __launch_bounds__(256) void k_append(Glob* g, int b, int nOld, int nNew, const double* depthSel, const double* depth2, double depthDefault,
    double pointVar, int cap, const double* bearings /* filter b */, const int* perm, double* p0, double* Q, double* lmc, int* errflag, T* S,
    long long sigmaStride, int ld) {
    const int nvo = kLm0 + 3 * nOld, nvn = kLm0 + 3 * (nOld + nNew);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    // (the bearing of this thread's new landmark -- a permutation entry, then three values behind it -- is requested BEFORE the median
    // selection: two dependent round trips that used to come after it)
    double yb[3] = {0.0, 0.0, 1.0};
    if (tid < nNew) {
        const int i = nOld + tid;
        const double* y = bearings + 3 * (perm ? perm[(long long)b * cap + i] : i);
        yb[0] = y[0]; yb[1] = y[1]; yb[2] = y[2];
    }
    __shared__ double sDepth;
    __shared__ double sD2[kMedianInAppend];
    if (nOld > 0 && !depthSel) {
        // the squared depths of the current estimate (k_probe's expression), once per workgroup into LDS -- the launch no longer needs a probe
        // launch in front of it (depth2 == nullptr; round 5)
        if (depth2) {
            for (int i = threadIdx.x; i < nOld; i += blockDim.x) sD2[i] = depth2[(long long)b * cap + i];
            __syncthreads();
        } else {
            for (int i = threadIdx.x; i < nOld; i += blockDim.x) {
                const double* P = p0 + (long long)b * 3 * cap;
                const double* q = Q + (long long)b * 5 * cap;
                const quat Qq = quat{q[i], q[cap + i], q[2 * cap + i], q[3 * cap + i]};
                const d3 qhat = scl(1.0 / q[4 * cap + i], qrot(qinv(Qq), mk3(P[i], P[cap + i], P[2 * cap + i])));
                sD2[i] = dot3(qhat, qhat);
            }
            __syncthreads();
        }
        const double* d = sD2;
        for (int i = threadIdx.x; i < nOld; i += blockDim.x) {
            const double di = d[i];
            int r0 = 0, r1 = 0, r2 = 0, r3 = 0;  // (four independent counters: the loads of a trip are in flight together)
            int j = 0;
            for (; j + 3 < nOld; j += 4) {
                const double d0 = d[j], d1 = d[j + 1], d2 = d[j + 2], d3 = d[j + 3];
                r0 += (d0 < di) || (d0 == di && j < i);
                r1 += (d1 < di) || (d1 == di && j + 1 < i);
                r2 += (d2 < di) || (d2 == di && j + 2 < i);
                r3 += (d3 < di) || (d3 == di && j + 3 < i);
            }
            for (; j < nOld; ++j) r0 += (d[j] < di) || (d[j] == di && j < i);
            if (r0 + r1 + r2 + r3 == nOld / 2) sDepth = sqrt(di);
        }
        __syncthreads();
    }
    const double depth = nOld > 0 ? (depthSel ? depthSel[b] : sDepth) : depthDefault;  // (:361-366)
    T* Sb = S + (long long)b * sigmaStride;
    // new rows (all columns) and new columns (old rows)
    const long long total = (long long)(nvn - nvo) * nvn + (long long)nvo * (nvn - nvo);
    for (long long e = tid; e < total; e += nth) {
        int R, Cc;
        if (e < (long long)(nvn - nvo) * nvn) {
            R = nvo + (int)(e / nvn);
            Cc = (int)(e % nvn);
        } else {
            const long long f = e - (long long)(nvn - nvo) * nvn;
            R = (int)(f / (nvn - nvo));
            Cc = nvo + (int)(f % (nvn - nvo));
        }
        Sb[(long long)R * ld + Cc] = (R == Cc) ? (T)pointVar : (T)0;
    }
    for (int j = tid; j < nNew; j += nth) {
        const int i = nOld + j;
        double yv[3] = {yb[0], yb[1], yb[2]};
        if (j != tid) {  // (more new landmarks than threads in the launch: cannot happen with its grid, kept for safety)
            const double* yp = bearings + 3 * (perm ? perm[(long long)b * cap + i] : i);
            yv[0] = yp[0]; yv[1] = yp[1]; yv[2] = yp[2];
        }
        const double* y = yv;
        // The origin landmark as it is STORED is what its constants must come from: a restored filter recomputes them from p0
        // (k_restore_constants) and has to continue bitwise (tests/test_replay.py).  Without the barrier the compiler may fuse y * depth
        // into the first operations of landmarkConstants -- it did, after an unrelated change of this kernel's shape.
        double px = y[0] * depth, py = y[1] * depth, pz = y[2] * depth;
#if defined(__HIP_DEVICE_COMPILE__)
        __asm__ volatile("" : "+v"(px), "+v"(py), "+v"(pz));
#endif
        p0[((long long)b * 3 + 0) * cap + i] = px;
        p0[((long long)b * 3 + 1) * cap + i] = py;
        p0[((long long)b * 3 + 2) * cap + i] = pz;
        Q[((long long)b * 5 + 0) * cap + i] = 1.0;
        Q[((long long)b * 5 + 1) * cap + i] = 0.0;
        Q[((long long)b * 5 + 2) * cap + i] = 0.0;
        Q[((long long)b * 5 + 3) * cap + i] = 0.0;
        Q[((long long)b * 5 + 4) * cap + i] = 1.0;
        double cst[15];
        int bad = 0;
        landmarkConstants(mk3(px, py, pz), cst, &bad);
        for (int c = 0; c < 15; ++c) lmc[((long long)b * 15 + c) * cap + i] = cst[c];
        if (bad && errflag) atomicOr(errflag, 16);
    }
    if (tid == 0) g[b].N = nOld + nNew;
}

// ------------------------------------------------------------------------------------------------
// k_edit (round 5): everything a vision frame does to the landmark SET in one launch, decided on the device --
//   removeOldLandmarks (VIOFilter.cpp:393-419: the host's keep list, it owns the ids), removeOutliers (:429-443: the chord of every kept
//   landmark against its measured bearing, here), addNewLandmarks (:345-391: at the median depth of what is left).
// Before it the same frame took up to five launches and two uploads (compaction, probe, median, append), and a frame whose gate tripped
// ran a discarded update launch and was redone from the host.  Now the outliers are compacted away right here and the update that
// follows in the stream sees the final set (N, the permutation into the measurement and the per-landmark arrays are all device-side);
// the host learns which ids went when it next touches the handle (chordOut / gateFlag in pinned memory: bookkeeping, nothing relaunched).
//
// in[b * cap + j]            map: old index of kept landmark j (ascending; the identity if the filter lost nothing)
// in[(B + b) * cap + i]      perm: measurement entry of landmark i of the order [kept ..., new ...]
// in[2 * B * cap + 4 * b]    {kept count, new count, gate armed for this filter, -}
// hostFlip: some filter of the batch lost a landmark, the host already counts on Sigma being in the OTHER buffer afterwards (the ping-pong
// parity is shared by the batch).  A filter whose only removals are outliers -- the host cannot know -- compacts into the other buffer,
// waits for its G workgroups (a counting barrier: the launch is sized to be co-resident) and copies the result back.
// grid = (G, B), block = 256 -- G any number when hostFlip is set (nobody waits for anybody then), at most the co-resident count otherwise;
// at most kEditMax kept landmarks per filter (the host falls back to the separate launches beyond).
constexpr int kEditMax = 1024;
constexpr int kEditSafeN = 59;  // from here on ceil((3 N + 5) / 64) > ceil(2 N / 64): the E-chain is the longer one whatever N is
struct EditArgs {
    Glob* g;
    const int* in;
    int* permOut;           // [B][cap]: the permutation the update will use
    int B, cap;
    const double* bearings;
    long long bearStride;
    double gateThr;
    int* gateFlag;          // pinned [B]: raised if the filter lost an outlier
    double* chordOut;       // pinned [B][cap]: chord of kept landmark j (gate armed)
    double depthDefault, pointVar;
    double *p0, *Q, *lmc;
    int* errflag;
    const void* Scur;
    void* Soth;
    long long sigmaStride;
    int ld, hostFlip;
    int* bar;               // [B][4] barrier arrivals, barrier generation, first-phase arrivals, - (zero-initialised once)
};
template <typename T>
__global__ __launch_bounds__(256) void k_edit(EditArgs a) {
    const int b = blockIdx.y, w = blockIdx.x, G = gridDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cap = a.cap;
    const int* mp = a.in + (long long)b * cap;
    const int* pm = a.in + (long long)(a.B + b) * cap;
    const int* cnt = a.in + (long long)2 * a.B * cap + 4 * b;
    // (the first trip's map and permutation entries are requested together with the counts: one round trip less on the launch's one chain)
    const int o0 = mp[min(tid, cap - 1)], k0 = pm[min(tid, cap - 1)];
    const int nK = min(cnt[0], kEditMax), nNew = cnt[1];
    const bool gate = cnt[2] != 0;
    const double* P = a.p0 + (long long)b * 3 * cap;
    const double* q = a.Q + (long long)b * 5 * cap;
    __shared__ double sD2[kEditMax];
    __shared__ int sOld[kEditMax];   // old index of FINAL landmark f
    __shared__ int sFin[kEditMax];   // kept index j of final landmark f
    __shared__ unsigned long long sMask[kEditMax / 64];
    __shared__ int sPre[kEditMax / 64 + 1];
    __shared__ double sDepth;
    // ---- every workgroup: depth and chord of the kept landmarks, the final keep list (k_probe's expressions)
    for (int base = 0; base < nK; base += 256) {
        const int j = base + tid;
        bool keep = false;
        if (j < nK) {
            const int o = base == 0 ? o0 : mp[j];
            const int k = base == 0 ? k0 : pm[j];
            double yv[3] = {0.0, 0.0, 1.0};
            if (gate && k >= 0) {
                const double* y = a.bearings + (long long)b * a.bearStride + 3 * k;
                yv[0] = y[0]; yv[1] = y[1]; yv[2] = y[2];
            }
            const quat Qq = quat{q[o], q[cap + o], q[2 * cap + o], q[3 * cap + o]};
            const d3 qhat = scl(1.0 / q[4 * cap + o], qrot(qinv(Qq), mk3(P[o], P[cap + o], P[2 * cap + o])));
            sD2[j] = dot3(qhat, qhat);
            double ch = 0.0;
            if (gate) {
                if (k >= 0) {
                    const double* y = yv;
                    ch = nrm3(sub(mk3(y[0], y[1], y[2]), unit3(qhat)));
                }
                if (w == 0 && a.chordOut) a.chordOut[(long long)b * cap + j] = ch;
            }
            keep = !(gate && ch > a.gateThr);
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) sMask[(base >> 6) + wv] = m;
    }
    __syncthreads();
    const int nChunks = (nK + 63) >> 6;
    if (tid == 0) {
        int acc = 0;
        for (int c = 0; c < nChunks; ++c) {
            sPre[c] = acc;
            acc += __popcll(sMask[c]);
        }
        sPre[nChunks] = acc;
    }
    __syncthreads();
    const int nF = nChunks ? sPre[nChunks] : 0;
    for (int j = tid; j < nK; j += 256) {
        const unsigned long long m = sMask[j >> 6];
        if ((m >> (j & 63)) & 1) {
            const int f = sPre[j >> 6] + __popcll(m & ((1ull << (j & 63)) - 1));
            sFin[f] = j;
            sOld[f] = mp[j];
        }
    }
    __syncthreads();
    // The per-landmark arrays are edited IN PLACE, and every workgroup has just read them: that work goes to the workgroup that finishes
    // this phase LAST (a counter, nobody waits), at the end of its share of Sigma.
    // (a frame that edits nothing in place -- the gate armed, nothing tripped, no new landmark -- skips the counter's round trip)
    const bool tripped = nF != nK;
    const bool moved = a.hostFlip || tripped;
    __shared__ int sLast;
    if (tid == 0) {
        if (moved || nNew > 0) {
            int* arr = a.bar + 4 * b + 2;
            const int old = __hip_atomic_fetch_add(arr, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            sLast = old == G - 1;
            if (old == G - 1) __hip_atomic_store(arr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            sLast = w == 0;
        }
    }
    __syncthreads();
    const bool last = sLast != 0;
    const int nvF = kLm0 + 3 * nF, nvn = kLm0 + 3 * (nF + nNew);
    const T* src = static_cast<const T*>(a.Scur) + (long long)b * a.sigmaStride;
    T* const cur = const_cast<T*>(src);
    T* dst = moved ? static_cast<T*>(a.Soth) + (long long)b * a.sigmaStride : cur;
    const int ld = a.ld;
    if (w == 0) {
        int* po = a.permOut + (long long)b * cap;
        for (int f = tid; f < nF; f += 256) po[f] = pm[sFin[f]];
        for (int t = tid; t < nNew; t += 256) po[nF + t] = pm[nK + t];
    }
    // ---- Sigma: the kept rows / columns into the other buffer
    if (moved) {
        // (four rows per trip: their loads are in flight together -- a load behind a store waits for it, the compiler cannot know they do not alias)
        for (int Cc = tid; Cc < nvF; Cc += 256) {
            const int Cs = (Cc < kLm0) ? Cc : kLm0 + 3 * sOld[(Cc - kLm0) / 3] + (Cc - kLm0) % 3;
            for (int R0 = w; R0 < nvF; R0 += 4 * G) {
                T v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int R = min(R0 + u * G, nvF - 1);
                    const int Rs = (R < kLm0) ? R : kLm0 + 3 * sOld[(R - kLm0) / 3] + (R - kLm0) % 3;
                    v[u] = src[(long long)Rs * ld + Cs];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (R0 + u * G < nvF) dst[(long long)(R0 + u * G) * ld + Cc] = v[u];
            }
        }
    }
    // ---- Sigma: rows and columns of the new landmarks (zero cross terms, initialPointVariance on the diagonal, :367-390)
    {
        const long long total = (long long)(nvn - nvF) * nvn + (long long)nvF * (nvn - nvF);
        for (long long e = (long long)w * 256 + tid; e < total; e += (long long)G * 256) {
            int R, Cc;
            if (e < (long long)(nvn - nvF) * nvn) {
                R = nvF + (int)(e / nvn);
                Cc = (int)(e % nvn);
            } else {
                const long long f = e - (long long)(nvn - nvF) * nvn;
                R = (int)(f / (nvn - nvF));
                Cc = nvF + (int)(f % (nvn - nvF));
            }
            dst[(long long)R * ld + Cc] = (R == Cc) ? (T)a.pointVar : (T)0;
        }
    }
    // ---- per-landmark arrays (the workgroup that left the first phase last): compaction in place, then the new landmarks
    if (last) {
        if (moved) {
            // (in place, 256 landmarks per pass: a record moves DOWN or stays, so a pass only reads what no earlier pass has written, and inside
            // a pass every load has returned -- the barrier drains them -- before the first store; k_compact's scratch copy, a round trip, is not
            // needed with one workgroup)
            for (int f0 = 0; f0 < nF; f0 += 256) {
                const int f = f0 + tid;
                double v[kLmRec];
                if (f < nF) {
                    const int o = sOld[f];
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = a.p0[((long long)b * 3 + c) * cap + o];
#pragma unroll
                    for (int c = 0; c < 5; ++c) v[3 + c] = a.Q[((long long)b * 5 + c) * cap + o];
#pragma unroll
                    for (int c = 0; c < 15; ++c) v[8 + c] = a.lmc[((long long)b * 15 + c) * cap + o];
                }
                __syncthreads();
                if (f < nF) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) a.p0[((long long)b * 3 + c) * cap + f] = v[c];
#pragma unroll
                    for (int c = 0; c < 5; ++c) a.Q[((long long)b * 5 + c) * cap + f] = v[3 + c];
#pragma unroll
                    for (int c = 0; c < 15; ++c) a.lmc[((long long)b * 15 + c) * cap + f] = v[8 + c];
                }
            }
        }
        if (nNew > 0) {
            // median scene depth of what is left (:357-366): k_append's rank counting over the final order
            if (nF > 0) {
                for (int i = tid; i < nF; i += 256) {
                    const double di = sD2[sFin[i]];
                    int rank = 0;
                    for (int j = 0; j < nF; ++j) {
                        const double dj = sD2[sFin[j]];
                        rank += (dj < di) || (dj == di && j < i);
                    }
                    if (rank == nF / 2) sDepth = sqrt(di);
                }
            }
            __syncthreads();
            const double depth = nF > 0 ? sDepth : a.depthDefault;
            for (int t = tid; t < nNew; t += 256) {
                const int i = nF + t;
                const double* y = a.bearings + (long long)b * a.bearStride + 3 * pm[nK + t];
                // (the origin landmark as it is STORED is what its constants come from: see k_append)
                double px = y[0] * depth, py = y[1] * depth, pz = y[2] * depth;
#if defined(__HIP_DEVICE_COMPILE__)
                __asm__ volatile("" : "+v"(px), "+v"(py), "+v"(pz));
#endif
                a.p0[((long long)b * 3 + 0) * cap + i] = px;
                a.p0[((long long)b * 3 + 1) * cap + i] = py;
                a.p0[((long long)b * 3 + 2) * cap + i] = pz;
                a.Q[((long long)b * 5 + 0) * cap + i] = 1.0;
                a.Q[((long long)b * 5 + 1) * cap + i] = 0.0;
                a.Q[((long long)b * 5 + 2) * cap + i] = 0.0;
                a.Q[((long long)b * 5 + 3) * cap + i] = 0.0;
                a.Q[((long long)b * 5 + 4) * cap + i] = 1.0;
                double cst[15];
                int bad = 0;
                landmarkConstants(mk3(px, py, pz), cst, &bad);
                for (int c = 0; c < 15; ++c) a.lmc[((long long)b * 15 + c) * cap + i] = cst[c];
                if (bad && a.errflag) atomicOr(a.errflag, 16);
            }
        }
        if (tid == 0) {
            a.g[b].N = nF + nNew;
            if (tripped && a.gateFlag) {
                // The update behind this launch was shaped for the host's count.  Below kEditSafeN landmarks the two chains of a filter can
                // be equally long, which needs another launch shape: if the outliers took the filter there, its update is switched off and
                // the host runs it when it looks at the flag (2).
                const bool defer = nF + nNew < kEditSafeN;
                if (defer) a.g[b].updateOk = 0;
                a.gateFlag[b] = defer ? 2 : 1;
            }
        }
    }
    // ---- outliers only: back into the buffer the host counts on
    if (tripped && !a.hostFlip) {
        __threadfence();
        __syncthreads();
        __shared__ int sOk;
        if (tid == 0) {
            int* arr = a.bar + 4 * b;
            const int gen = __hip_atomic_load(arr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int ok = 1;
            if (__hip_atomic_fetch_add(arr, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
                __hip_atomic_store(arr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(arr + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                long long spins = 0;
                while (__hip_atomic_load(arr + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1LL << 24)) {  // (never co-resident after all: seconds, not a hang)
                        ok = 0;
                        break;
                    }
                }
            }
            sOk = ok;
        }
        __syncthreads();
        __threadfence();
        if (!sOk) {
            if (tid == 0 && a.errflag) atomicOr(a.errflag, kHoErrTimeout);
            return;
        }
        for (int Cc = tid; Cc < nvn; Cc += 256)
            for (int R0 = w; R0 < nvn; R0 += 4 * G) {
                T v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = dst[(long long)min(R0 + u * G, nvn - 1) * ld + Cc];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (R0 + u * G < nvn) cur[(long long)(R0 + u * G) * ld + Cc] = v[u];
            }
    }
}

// Restore path: per-landmark constants and cached pose constants recomputed on the device after eqf_set_state.
__global__ void k_restore_constants(Glob* g, int b, const double* p0, double* lmc, int cap, int* errflag) {
    Glob& s = g[b];
    int bad = 0;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = tid; i < s.N; i += gridDim.x * blockDim.x) {
        double cst[15];
        landmarkConstants(mk3(p0[((long long)b * 3 + 0) * cap + i], p0[((long long)b * 3 + 1) * cap + i], p0[((long long)b * 3 + 2) * cap + i]), cst, &bad);
        for (int c = 0; c < 15; ++c) lmc[((long long)b * 15 + c) * cap + i] = cst[c];
    }
    if (tid == 0 && s.initialised) {
        double e0[3], cd[6], ci[6];
        poseConstants(quat{s.P0q[0], s.P0q[1], s.P0q[2], s.P0q[3]}, e0, cd, ci, &bad);
        for (int i = 0; i < 3; ++i) s.eta0[i] = e0[i];
        for (int i = 0; i < 6; ++i) {
            s.cDiff[i] = cd[i];
            s.cInv[i] = ci[i];
        }
    }
    if (bad && errflag) atomicOr(errflag, 32);
}

// stateEstimate = stateGroupAction(X, xi0) (VIOFilter.cpp:304, VIOGroup.cpp:23-45): out[b] = q(4) x(3) v(3) p(3N)
__global__ void k_state_estimate(const Glob* g, int b, const double* p0, const double* Q, int cap, double* out) {
    const Glob& s = g[b];
    const se3 P0 = se3{quat{s.P0q[0], s.P0q[1], s.P0q[2], s.P0q[3]}, mk3(s.P0x[0], s.P0x[1], s.P0x[2])};
    const se3 A = se3{quat{s.Aq[0], s.Aq[1], s.Aq[2], s.Aq[3]}, mk3(s.Ax[0], s.Ax[1], s.Ax[2])};
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid == 0) {
        const se3 P = se3mul(P0, A);
        const d3 v = qrot(qinv(A.q), mk3(s.v0[0] - s.w[0], s.v0[1] - s.w[1], s.v0[2] - s.w[2]));
        out[0] = P.q.w; out[1] = P.q.x; out[2] = P.q.y; out[3] = P.q.z;
        out[4] = P.x.x; out[5] = P.x.y; out[6] = P.x.z;
        out[7] = v.x; out[8] = v.y; out[9] = v.z;
    }
    for (int i = tid; i < s.N; i += gridDim.x * blockDim.x) {
        const double* P = p0 + (long long)b * 3 * cap;
        const double* q = Q + (long long)b * 5 * cap;
        const quat Qq = quat{q[i], q[cap + i], q[2 * cap + i], q[3 * cap + i]};
        const d3 qh = scl(1.0 / q[4 * cap + i], qrot(qinv(Qq), mk3(P[i], P[cap + i], P[2 * cap + i])));
        out[10 + 3 * i] = qh.x; out[11 + 3 * i] = qh.y; out[12 + 3 * i] = qh.z;
    }
}

// Sigma <-> reference index map (drop / insert the pad row+column 11), fp64 on the host side.
template <typename T>
__global__ void k_sigma_export(const T* S, int ld, int n /* 11 + 3N */, double* out, int ldo) {
    const int R = blockIdx.y;
    const int Rs = R < kBase ? R : R + 1;
    for (int Cc = blockIdx.x * blockDim.x + threadIdx.x; Cc < n; Cc += gridDim.x * blockDim.x) {
        const int Cs = Cc < kBase ? Cc : Cc + 1;
        out[(long long)R * ldo + Cc] = (double)S[(long long)Rs * ld + Cs];
    }
}
template <typename T>
__global__ void k_sigma_import(T* S, int ld, int n, const double* in, int ldi) {
    const int R = blockIdx.y;  // internal row in [0, n + 1)
    for (int Cc = blockIdx.x * blockDim.x + threadIdx.x; Cc < n + 1; Cc += gridDim.x * blockDim.x) {
        T v = 0;
        if (R != kBase && Cc != kBase) {
            const int Rr = R < kBase ? R : R - 1, Cr = Cc < kBase ? Cc : Cc - 1;
            v = (T)in[(long long)Rr * ldi + Cr];
        }
        S[(long long)R * ld + Cc] = v;
    }
}

}  // namespace eqf
