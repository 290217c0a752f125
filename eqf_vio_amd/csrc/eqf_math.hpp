// Device-side fp64 Lie-group / sphere-chart helpers for the EqF kernels (gfx950).
//
// The O(N) geometric state (X, xi0, lifts, chart differentials) is kept in fp64 on the device: it is
// a few hundred flops per landmark and feeds the linearisation blocks, so there is nothing to gain from
// lower precision.  Formulas follow the reference's libs/core (SO3.cpp, SE3.cpp, SOT3.cpp) and
// VIOState.cpp:199-251; rotations are stored as quaternions like the reference does (SO3.cpp:25) and
// converted with Eigen's published quaternion<->matrix algorithms, so the fp64 oracle and the device
// agree to rounding.
#pragma once
#include <hip/hip_runtime.h>

#define EQF_DI __host__ __device__ __forceinline__

namespace eqf {

constexpr double kGravity = 9.81;  // eqf_vio/include/eqf_vio/IMUVelocity.h:22

struct d3 {
    double x, y, z;
};
// a0 b0 + a1 b1 + a2 b2 with the roundings pinned (one product, two fused multiply-adds): the rows of C Sigma and the blocks of S are formed
// by k_update_prep64's landmark waves OR by the burst's block kernel (BurstArgs::csOut), and the two must agree bit for bit.
EQF_DI double dot3(double a0, double b0, double a1, double b1, double a2, double b2) { return fma(a2, b2, fma(a1, b1, a0 * b0)); }
EQF_DI d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
EQF_DI d3 add(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
EQF_DI d3 sub(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
EQF_DI d3 neg(d3 a) { return d3{-a.x, -a.y, -a.z}; }
EQF_DI d3 scl(double c, d3 a) { return d3{c * a.x, c * a.y, c * a.z}; }
EQF_DI double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
EQF_DI d3 crs(d3 a, d3 b) { return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
EQF_DI double nrm3(d3 a) { return sqrt(dot3(a, a)); }
EQF_DI d3 unit3(d3 a) { return scl(1.0 / nrm3(a), a); }
EQF_DI double comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// 3x3 row-major
struct m33 {
    double a[9];
};
EQF_DI m33 eye3() { return m33{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
EQF_DI m33 mul33(const m33& A, const m33& B) {
    m33 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
    return C;
}
EQF_DI m33 mulT33(const m33& A, const m33& B) {  // A * B^T
    m33 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C.a[3 * i + j] = A.a[3 * i] * B.a[3 * j] + A.a[3 * i + 1] * B.a[3 * j + 1] + A.a[3 * i + 2] * B.a[3 * j + 2];
    return C;
}
EQF_DI m33 tr33(const m33& A) { return m33{{A.a[0], A.a[3], A.a[6], A.a[1], A.a[4], A.a[7], A.a[2], A.a[5], A.a[8]}}; }
EQF_DI d3 mv33(const m33& A, d3 v) {
    return d3{A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z, A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z,
        A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z};
}
EQF_DI d3 mtv33(const m33& A, d3 v) {  // A^T v
    return d3{A.a[0] * v.x + A.a[3] * v.y + A.a[6] * v.z, A.a[1] * v.x + A.a[4] * v.y + A.a[7] * v.z,
        A.a[2] * v.x + A.a[5] * v.y + A.a[8] * v.z};
}
EQF_DI m33 add33(const m33& A, const m33& B) {
    m33 C;
#pragma unroll
    for (int i = 0; i < 9; ++i) C.a[i] = A.a[i] + B.a[i];
    return C;
}
EQF_DI m33 scl33(double c, const m33& A) {
    m33 C;
#pragma unroll
    for (int i = 0; i < 9; ++i) C.a[i] = c * A.a[i];
    return C;
}
EQF_DI m33 skew3(d3 v) { return m33{{0, -v.z, v.y, v.z, 0, -v.x, -v.y, v.x, 0}}; }  // SO3.cpp:110-114
EQF_DI m33 outer3(d3 a, d3 b) {
    return m33{{a.x * b.x, a.x * b.y, a.x * b.z, a.y * b.x, a.y * b.y, a.y * b.z, a.z * b.x, a.z * b.y, a.z * b.z}};
}

// quaternion (w, x, y, z), Eigen semantics
struct quat {
    double w, x, y, z;
};
EQF_DI m33 q2m(quat q) {  // Eigen toRotationMatrix (SO3.cpp:94)
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    return m33{{1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx,
        1 - (txx + tyy)}};
}
EQF_DI quat m2q(const m33& M) {  // Eigen matrix -> quaternion (SO3.cpp:100)
    const double m00 = M.a[0], m11 = M.a[4], m22 = M.a[8];
    double t = m00 + m11 + m22;
    quat q;
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (M.a[7] - M.a[5]) * t;
        q.y = (M.a[2] - M.a[6]) * t;
        q.z = (M.a[3] - M.a[1]) * t;
    } else if (m00 >= m11 && m00 >= m22) {  // i = 0, j = 1, k = 2
        t = sqrt(m00 - m11 - m22 + 1.0);
        q.x = 0.5 * t;
        t = 0.5 / t;
        q.w = (M.a[7] - M.a[5]) * t;
        q.y = (M.a[3] + M.a[1]) * t;
        q.z = (M.a[6] + M.a[2]) * t;
    } else if (m11 > m00 && m11 >= m22) {  // i = 1, j = 2, k = 0
        t = sqrt(m11 - m22 - m00 + 1.0);
        q.y = 0.5 * t;
        t = 0.5 / t;
        q.w = (M.a[2] - M.a[6]) * t;
        q.z = (M.a[7] + M.a[5]) * t;
        q.x = (M.a[1] + M.a[3]) * t;
    } else {  // i = 2, j = 0, k = 1
        t = sqrt(m22 - m00 - m11 + 1.0);
        q.z = 0.5 * t;
        t = 0.5 / t;
        q.w = (M.a[3] - M.a[1]) * t;
        q.x = (M.a[2] + M.a[6]) * t;
        q.y = (M.a[5] + M.a[7]) * t;
    }
    return q;
}
EQF_DI quat qmul(quat a, quat b) {  // SO3.cpp:76
    return quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
        a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
EQF_DI quat qinv(quat q) {  // conjugate / squaredNorm (SO3.cpp:82)
    const double r = 1.0 / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    return quat{q.w * r, -q.x * r, -q.y * r, -q.z * r};
}
EQF_DI d3 qrot(quat q, d3 v) {  // Eigen _transformVector (SO3.cpp:66)
    const d3 u = mk3(q.x, q.y, q.z);
    d3 uv = crs(u, v);
    uv = add(uv, uv);
    return add(add(v, scl(q.w, uv)), crs(u, uv));
}

// SO3FromVectors(origin, dest) for UNIT inputs (SO3.cpp:155-167).  *bad is set when |1+c| <= 1e-8
// (the reference throws std::domain_error there).
EQF_DI m33 rotFromUnitVectors(d3 o, d3 d, int* bad) {
    const d3 v = crs(o, d);
    const double c = dot3(o, d);
    if (fabs(1 + c) <= 1e-8) *bad = 1;
    const m33 vx = skew3(v);
    return add33(eye3(), add33(vx, scl33(1.0 / (1.0 + c), mul33(vx, vx))));
}
EQF_DI quat so3FromVectors(d3 origin, d3 dest, int* bad) { return m2q(rotFromUnitVectors(unit3(origin), unit3(dest), bad)); }

// Coefficients of the SO(3)/SE(3) exponentials (SO3.cpp:122-140, SE3.cpp:139-164):
//   A = sin(th)/th,  B = (1-cos th)/th^2,  C = (1-A)/th^2      as functions of t = th^2.
// For th^2 < 0.25 (every IMU- or innovation-sized increment) they come from their Maclaurin series in t: no sqrt,
// no sin/cos (fp64 sin/cos cost thousands of cycles on one lane) and no cancellation -- the reference's
// (1-cos th)/th^2 loses ~1e-16/th^2 relative accuracy for small th; the series is exact to rounding.  Larger angles
// use the reference's closed forms.
EQF_DI void expCoefficients(double t, double* A, double* B, double* C) {
    if (t < 0.25) {
        // A = sum (-t)^k/(2k+1)!,  B = sum (-t)^k/(2k+2)!,  C = sum (-t)^k/(2k+3)!   (k = 0..9: remainder < 1e-20)
        double a = 1.0 / 121645100408832000.0, b = 1.0 / 2432902008176640000.0, c = 1.0 / 51090942171709440000.0;
        const double ia[9] = {1.0 / 355687428096000.0, 1.0 / 1307674368000.0, 1.0 / 6227020800.0, 1.0 / 39916800.0,
            1.0 / 362880.0, 1.0 / 5040.0, 1.0 / 120.0, 1.0 / 6.0, 1.0};
        const double ib[9] = {1.0 / 6402373705728000.0, 1.0 / 20922789888000.0, 1.0 / 87178291200.0, 1.0 / 479001600.0,
            1.0 / 3628800.0, 1.0 / 40320.0, 1.0 / 720.0, 1.0 / 24.0, 0.5};
        const double ic[9] = {1.0 / 121645100408832000.0, 1.0 / 355687428096000.0, 1.0 / 1307674368000.0, 1.0 / 6227020800.0,
            1.0 / 39916800.0, 1.0 / 362880.0, 1.0 / 5040.0, 1.0 / 120.0, 1.0 / 6.0};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            a = ia[k] - t * a;
            b = ib[k] - t * b;
            c = ic[k] - t * c;
        }
        *A = a;
        *B = b;
        *C = c;
    } else {
        const double th = sqrt(t);
        *A = sin(th) / th;
        *B = (1 - cos(th)) / t;
        *C = (1 - *A) / t;
    }
}

EQF_DI quat so3Exp(d3 w) {  // SO3.cpp:122-140
    double A, B, C;
    expCoefficients(dot3(w, w), &A, &B, &C);
    const m33 wx = skew3(w);
    return m2q(add33(eye3(), add33(scl33(A, wx), scl33(B, mul33(wx, wx)))));
}

struct se3 {
    quat q;
    d3 x;
};
EQF_DI se3 se3mul(se3 a, se3 b) { return se3{qmul(a.q, b.q), add(a.x, qrot(a.q, b.x))}; }  // SE3.cpp:71-76
EQF_DI se3 se3inv(se3 a) {                                                                    // SE3.cpp:80-83
    const quat qi = qinv(a.q);
    return se3{qi, neg(qrot(qi, a.x))};
}
EQF_DI d3 se3app(se3 a, d3 p) { return add(qrot(a.q, p), a.x); }  // SE3.cpp:63
EQF_DI se3 se3Exp(d3 w, d3 v) {                                      // SE3.cpp:139-164
    double A, B, C;
    expCoefficients(dot3(w, w), &A, &B, &C);
    const m33 wx = skew3(w);
    const m33 wx2 = mul33(wx, wx);
    const m33 R = add33(eye3(), add33(scl33(A, wx), scl33(B, wx2)));
    const m33 V = add33(eye3(), add33(scl33(B, wx), scl33(C, wx2)));
    return se3{m2q(R), mv33(V, v)};
}
// Ad(T) (w; v) = (R w ; x^ R w + R v)   (SE3.cpp:95-103)
EQF_DI void se3AdjointApply(se3 T, d3 w, d3 v, d3* ow, d3* ov) {
    const m33 R = q2m(T.q);
    const d3 Rw = mv33(R, w);
    *ow = Rw;
    *ov = add(crs(T.x, Rw), mv33(R, v));
}

// Sphere chart frame: the rotation R_s(pole) = SO3FromVectors(-pole, e3) AS THE REFERENCE STORES IT,
// i.e. passed through matrix -> quaternion -> matrix (VIOState.cpp:231, SO3.cpp:100).
EQF_DI quat sphereRotQ(d3 pole, int* bad) { return so3FromVectors(neg(pole), mk3(0, 0, 1), bad); }

// stereoSphereChart(eta, pole) (VIOState.cpp:230-234, :199-204)
EQF_DI void stereoChart(d3 eta, d3 pole, double* y0, double* y1, int* bad) {
    const d3 r = qrot(sphereRotQ(pole, bad), eta);
    *y0 = r.x / (1 - r.z);
    *y1 = r.y / (1 - r.z);
}
// stereoSphereChartDiff(eta, pole) 2x3 (VIOState.cpp:242-246, :213-220); out row-major [6]
EQF_DI void stereoChartDiff(d3 eta, d3 pole, double* out, int* bad) {
    const quat q = sphereRotQ(pole, bad);
    const d3 r = qrot(q, eta);
    const m33 R = q2m(q);
    const double s = 1.0 / ((1 - r.z) * (1 - r.z));
    const double d00 = (1 - r.z) * s, d02 = r.x * s, d12 = r.y * s;  // rows of e3ProjectSphereDiff
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        out[j] = d00 * R.a[j] + d02 * R.a[6 + j];
        out[3 + j] = d00 * R.a[3 + j] + d12 * R.a[6 + j];
    }
}
// stereoSphereChartInvDiff(0, pole) 3x2 (VIOState.cpp:248-251, :222-228); out row-major [6]
EQF_DI void stereoChartInvDiffAtZero(d3 pole, double* out, int* bad) {
    const m33 R = q2m(qinv(sphereRotQ(pole, bad)));
    // e3ProjectSphereInvDiff(0) = 2 * [[1,0],[0,1],[0,0]]
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        out[2 * i] = 2.0 * R.a[3 * i];
        out[2 * i + 1] = 2.0 * R.a[3 * i + 1];
    }
}

}  // namespace eqf
