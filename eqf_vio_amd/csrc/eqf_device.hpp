// Device-resident filter state of the MI355X EqF path, shared by the kernels and the C ABI host code.
#pragma once
#include <hip/hip_runtime.h>

namespace eqf {

// Internal Sigma layout: the reference's 11 base coordinates (VIOFilter.cpp:54-57) are padded to 12 so
// that landmark blocks start 16-byte aligned and a 16-landmark tile is 48 contiguous values:
//   internal index = ref index            for ref index < 11
//                  = 12 + 3 i + c         for landmark i, component c   (ref index 11 + 3 i + c)
// Row / column 11 is structurally zero.  eqf_get_sigma / eqf_set_sigma translate.
constexpr int kBase = 11;
constexpr int kLm0 = 12;
constexpr int kTileLm = 16;            // landmarks per propagate tile edge
constexpr int kTile = 3 * kTileLm;     // 48 Sigma rows/cols per tile edge
constexpr int kNB = 32;                // Cholesky block size

__host__ __device__ inline int lmRow(int i) { return kLm0 + 3 * i; }

// Per-filter scalar state (fp64).  Double-buffered: a step reads buffer `cur` and writes the other.
struct Glob {
    double P0q[4], P0x[3];  // xi0.pose          (VIOState.h:52)
    double v0[3];           // xi0.velocity      (VIOState.h:53)
    double Aq[4], Ax[3];    // X.A               (VIOGroup.h:25)
    double w[3];            // X.w               (VIOGroup.h:26)
    double bias[6];         // inputBias         (VIOFilter.h:45)
    double curVel[6];       // currentVelocity   (VIOFilter.h:52)
    double accVel[6];       // accumulatedVelocity (VIOFilter.h:54)
    double accTime;         // accumulatedTime   (VIOFilter.h:55)
    double curTime;         // currentTime       (VIOFilter.h:51)
    int initialised;        // initialisedFlag   (VIOFilter.h:50)
    int N;                  // number of landmarks
    int updateOk;           // vision call: integrateUpToTime succeeded && initialised (VIOFilter.cpp:234-236)
    int pad_;
    // Functions of xi0.pose only, cached when the pose is set (xi0 never changes between landmark-set changes):
    double eta0[3];         // gravity direction R_P0^T e3                         (VIOState.cpp:90)
    double cDiff[6];        // stereoSphereChartDiff(eta0, eta0)       2x3         (EqFMatrices.cpp:364)
    double cInv[6];         // stereoSphereChartInvDiff(0, eta0)       3x2         (EqFMatrices.cpp:289)
};

// Tunables the kernels need (VIOFilterSettings.h:28-54), passed by value.
struct Params {
    double biasOmegaProcessVariance, biasAccelProcessVariance, gravityProcessVariance, velocityProcessVariance,
        pointProcessVariance, velOmegaVariance, velAccelVariance, measurementVariance, initialPointVariance;
    double camq[4], camx[3];
    int useInnovationLift, useDiscreteInnovationLift, useDiscreteVelocityLift;
    // constants of the camera offset, precomputed on the host with the same formulas the kernels use
    double RIC[9];          // matrix of cameraOffset.R
    double RICt[9];         // matrix of cameraOffset.R.inverse()      (EqFMatrices.cpp:371)
    double camIq[4], camIx[3];  // cameraOffset.inverse()               (SE3.cpp:80-83)
    double RcamI[9];        // matrix of cameraOffset.inverse().R       (for Ad(T_IC^-1), SE3.cpp:95-103)
};

// One IMU record: stamp, omega, accel, pad (IMUVelocity.h:24-37)
struct ImuRec {
    double stamp, w[3], a[3], pad_;
};

// Per-landmark SoA (fp64) of one filter: origin landmark p0 (xi0.bodyLandmarks[i].p) and Q_i = (q, a).
struct LmPtrs {
    double* p0;  // [3][cap]
    double* Q;   // [5][cap]: qw qx qy qz a
};

}  // namespace eqf
