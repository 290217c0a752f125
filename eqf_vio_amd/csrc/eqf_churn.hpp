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
