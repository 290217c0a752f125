// TEST INFRASTRUCTURE ONLY -- dependency-free fp64 C++17 restatement of the eqf_vio EqF hot path.
//
// This file is the checker and the timed CPU baseline ("port"), never the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.  It follows, in own
// code, pvangoor/eqf_vio (paths relative to /root/reference/eqf_vio) and executes the reference's
// DENSE operation sequence in the reference's evaluation order:
//   src/VIOFilter.cpp:120-209   processIMUData / integrateUpToTime  (Sigma <- T(P+BRB') + (F Sigma) F')
//   src/VIOFilter.cpp:232-302   processVisionData  (S, K = (Sigma C') S^-1, Sigma - (K C) Sigma)
//   src/VIOFilter.cpp:345-443   landmark bookkeeping
//   src/EqFMatrices.cpp:173-382 bundleLift, liftTotalSpaceInnovationDiscrete, A0, B, C0
//   src/VIOGroup.cpp, src/VIOState.cpp, src/VisionMeasurement.cpp, libs/core/src/{SO3,SE3,SOT3}.cpp
//
// PARITY UNPINNED at the filter level: the reference needs Eigen 3 + yaml-cpp (absent here) and its
// tests hold no golden vector for VIOFilter.  Pins: the reference's property tests restated in
// tests/test_oracle_properties.py, and agreement with the independent numpy restatement
// oracle/eqf_numpy.py (tests/test_oracle_cross.py).
//
// Third-party arithmetic (Eigen 3, unpinned): quaternion<->matrix, quaternion product/inverse/rotate
// follow Eigen/src/Geometry/Quaternion.h; MatrixXd::inverse() is restated as LU with partial
// pivoting; householderQr().solve() on the 4x4 system as Householder QR.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace {

constexpr double GRAVITY_CONSTANT = 9.81;  // include/eqf_vio/IMUVelocity.h:22
constexpr int SIGMA_BASE_SIZE = 11;        // include/eqf_vio/VIOFilter.h:28

// ------------------------------------------------------------------ small fixed-size algebra
struct V3 {
    double x = 0, y = 0, z = 0;
    double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double c, V3 a) { return {c * a.x, c * a.y, c * a.z}; }
inline V3 operator*(V3 a, double c) { return c * a; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) { return (1.0 / norm(a)) * a; }

struct M3 {
    double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    static M3 I() {
        M3 r;
        r.m[0][0] = r.m[1][1] = r.m[2][2] = 1;
        return r;
    }
};
inline M3 operator*(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
inline V3 operator*(const M3& a, V3 v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
        a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline M3 operator+(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
    return r;
}
inline M3 operator-(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] - b.m[i][j];
    return r;
}
inline M3 operator*(double c, const M3& a) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = c * a.m[i][j];
    return r;
}
inline M3 transpose(const M3& a) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
inline M3 outer(V3 a, V3 b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a[i] * b[j];
    return r;
}
inline M3 skew(V3 v) {  // libs/core/src/SO3.cpp:110-114
    M3 r;
    r.m[0][1] = -v.z; r.m[0][2] = v.y;
    r.m[1][0] = v.z;  r.m[1][2] = -v.x;
    r.m[2][0] = -v.y; r.m[2][1] = v.x;
    return r;
}
inline M3 inverse3(const M3& a) {  // Eigen Matrix3d::inverse(): cofactor formula (EqFMatrices.cpp:310)
    const double(*m)[3] = a.m;
    M3 c;
    c.m[0][0] = m[1][1] * m[2][2] - m[1][2] * m[2][1];
    c.m[0][1] = m[0][2] * m[2][1] - m[0][1] * m[2][2];
    c.m[0][2] = m[0][1] * m[1][2] - m[0][2] * m[1][1];
    c.m[1][0] = m[1][2] * m[2][0] - m[1][0] * m[2][2];
    c.m[1][1] = m[0][0] * m[2][2] - m[0][2] * m[2][0];
    c.m[1][2] = m[0][2] * m[1][0] - m[0][0] * m[1][2];
    c.m[2][0] = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    c.m[2][1] = m[0][1] * m[2][0] - m[0][0] * m[2][1];
    c.m[2][2] = m[0][0] * m[1][1] - m[0][1] * m[1][0];
    const double det = m[0][0] * c.m[0][0] + m[0][1] * c.m[1][0] + m[0][2] * c.m[2][0];
    return (1.0 / det) * c;
}

// ------------------------------------------------------------------ quaternion (Eigen semantics)
struct Quat {
    double w = 1, x = 0, y = 0, z = 0;
};
inline M3 toMatrix(const Quat& q) {  // QuaternionBase::toRotationMatrix  (SO3.cpp:94)
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 r;
    r.m[0][0] = 1 - (tyy + tzz); r.m[0][1] = txy - twz; r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz; r.m[1][1] = 1 - (txx + tzz); r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy; r.m[2][1] = tyz + twx; r.m[2][2] = 1 - (txx + tyy);
    return r;
}
inline Quat fromMatrix(const M3& a) {  // Eigen matrix -> quaternion (SO3.cpp:100)
    const double(*m)[3] = a.m;
    double t = m[0][0] + m[1][1] + m[2][2];
    double q[4];  // w, x, y, z
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m[2][1] - m[1][2]) * t;
        q[2] = (m[0][2] - m[2][0]) * t;
        q[3] = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[k][j] - m[j][k]) * t;
        q[1 + j] = (m[j][i] + m[i][j]) * t;
        q[1 + k] = (m[k][i] + m[i][k]) * t;
    }
    return {q[0], q[1], q[2], q[3]};
}
inline Quat operator*(const Quat& a, const Quat& b) {  // SO3.cpp:76
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
        a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Quat inverse(const Quat& q) {  // conjugate / squaredNorm (SO3.cpp:82)
    const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
inline V3 rotate(const Quat& q, V3 v) {  // _transformVector (SO3.cpp:66)
    const V3 u{q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + q.w * uv + cross(u, uv);
}

struct Antipodal : std::domain_error {
    Antipodal() : std::domain_error("The vectors cannot be exactly opposing.") {}
};

inline Quat SO3FromVectors(V3 origin, V3 dest) {  // SO3.cpp:155-167
    const V3 o = normalized(origin), d = normalized(dest);
    const V3 v = cross(o, d);
    const double c = dot(o, d);
    const M3 vx = skew(v);
    const M3 mat = M3::I() + (vx + (1 / (1 + c)) * (vx * vx));
    if (std::abs(1 + c) <= 1e-8) throw Antipodal();
    return fromMatrix(mat);
}
inline Quat SO3Exp(V3 w) {  // SO3.cpp:122-140
    const double th = norm(w);
    double A, B;
    if (std::abs(th) >= 1e-8) {
        A = std::sin(th) / th;
        B = (1 - std::cos(th)) / std::pow(th, 2);
    } else {
        A = 1.0;
        B = 0.5;
    }
    const M3 wx = skew(w);
    return fromMatrix(M3::I() + A * wx + B * (wx * wx));
}

struct SE3 {
    Quat q;
    V3 x;
    V3 apply(V3 p) const { return rotate(q, p) + x; }                                   // SE3.cpp:63
    SE3 operator*(const SE3& o) const { return {q * o.q, x + rotate(q, o.x)}; }          // SE3.cpp:71-76
    SE3 inv() const {                                                                    // SE3.cpp:80-83
        const Quat qi = inverse(q);
        return {qi, -rotate(qi, x)};
    }
};
inline SE3 SE3Exp(const double u[6]) {  // SE3.cpp:139-164
    const V3 w{u[0], u[1], u[2]}, v{u[3], u[4], u[5]};
    const double th = norm(w);
    double A, B, C;
    if (std::abs(th) >= 1e-12) {
        A = std::sin(th) / th;
        B = (1 - std::cos(th)) / std::pow(th, 2);
        C = (1 - A) / std::pow(th, 2);
    } else {
        A = 1.0; B = 0.5; C = 1.0 / 6.0;
    }
    const M3 wx = skew(w);
    const M3 wx2 = wx * wx;
    const M3 R = M3::I() + A * wx + B * wx2;
    const M3 V = M3::I() + B * wx + C * wx2;
    return {fromMatrix(R), V * v};
}
// Ad(T) applied to (w; v): [R w ; x^ R w + R v]  (SE3.cpp:95-103)
inline void adjointApply(const SE3& T, const double u[6], double out[6]) {
    const M3 R = toMatrix(T.q);
    const V3 Rw = R * V3{u[0], u[1], u[2]};
    const V3 Rv = R * V3{u[3], u[4], u[5]};
    const V3 lo = skew(T.x) * Rw + Rv;
    out[0] = Rw.x; out[1] = Rw.y; out[2] = Rw.z;
    out[3] = lo.x; out[4] = lo.y; out[5] = lo.z;
}

struct SOT3 {
    Quat q;
    double a = 1;
    SOT3 operator*(const SOT3& o) const { return {q * o.q, a * o.a}; }  // SOT3.cpp:77-82
    SOT3 inv() const { return {inverse(q), 1.0 / a}; }                  // SOT3.cpp:86-89
    V3 apply(V3 p) const { return a * rotate(q, p); }                   // SOT3.cpp:69
    M3 asMatrix3() const { return a * toMatrix(q); }                    // SOT3.cpp:107-110
};

// ------------------------------------------------------------------ sphere charts (VIOState.cpp:199-251)
struct M23 { double m[2][3]; };
struct M32 { double m[3][2]; };
inline void e3ProjectSphere(V3 eta, double y[2]) {
    y[0] = eta.x / (1 - eta.z);
    y[1] = eta.y / (1 - eta.z);
}
inline M23 e3ProjectSphereDiff(V3 eta) {  // :213-220
    // I23 * (I (1 - eta_z) + (eta - e3) e3^T) * (1 - eta_z)^-2
    M23 D{};
    const double s = std::pow(1 - eta.z, -2.0);
    D.m[0][0] = (1 - eta.z) * s; D.m[0][1] = 0; D.m[0][2] = eta.x * s;
    D.m[1][0] = 0; D.m[1][1] = (1 - eta.z) * s; D.m[1][2] = eta.y * s;
    return D;
}
inline M32 e3ProjectSphereInvDiff(const double y[2]) {  // :222-228
    M32 D{};
    const double n2 = y[0] * y[0] + y[1] * y[1];
    const double s = 2.0 * std::pow(n2 + 1.0, -2.0);
    D.m[0][0] = ((n2 + 1.0) - 2 * y[0] * y[0]) * s; D.m[0][1] = (-2 * y[0] * y[1]) * s;
    D.m[1][0] = (-2 * y[1] * y[0]) * s; D.m[1][1] = ((n2 + 1.0) - 2 * y[1] * y[1]) * s;
    D.m[2][0] = 2 * y[0] * s; D.m[2][1] = 2 * y[1] * s;
    return D;
}
inline Quat sphereRot(V3 pole) { return SO3FromVectors(-pole, V3{0, 0, 1}); }
inline void stereoSphereChart(V3 eta, V3 pole, double y[2]) { e3ProjectSphere(rotate(sphereRot(pole), eta), y); }  // :230-234
inline M23 stereoSphereChartDiff(V3 eta, V3 pole) {  // :242-246
    const Quat q = sphereRot(pole);
    const M23 D = e3ProjectSphereDiff(rotate(q, eta));
    const M3 R = toMatrix(q);
    M23 out{};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += D.m[i][k] * R.m[k][j];
            out.m[i][j] = s;
        }
    return out;
}
inline M32 stereoSphereChartInvDiff(const double y[2], V3 pole) {  // :248-251
    const M3 R = toMatrix(inverse(sphereRot(pole)));
    const M32 D = e3ProjectSphereInvDiff(y);
    M32 out{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += R.m[i][k] * D.m[k][j];
            out.m[i][j] = s;
        }
    return out;
}

// ------------------------------------------------------------------ dense matrices (row-major)
struct Mat {
    int r = 0, c = 0;
    std::vector<double> d;
    Mat() = default;
    Mat(int r_, int c_) : r(r_), c(c_), d(size_t(r_) * c_, 0.0) {}
    double& operator()(int i, int j) { return d[size_t(i) * c + j]; }
    double operator()(int i, int j) const { return d[size_t(i) * c + j]; }
    static Mat Identity(int n) {
        Mat m(n, n);
        for (int i = 0; i < n; ++i) m(i, i) = 1;
        return m;
    }
};
inline void setBlock(Mat& A, int i0, int j0, const M3& b, double s = 1.0) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A(i0 + i, j0 + j) = s * b.m[i][j];
}

// C = A * B (transB = false) or A * B^T (transB = true).  Single-thread, cache-blocked, with a packed-B
// 6x8 AVX2 register tile -- a stand-in for Eigen's GEBP kernel (Eigen is not available here), so that
// the timed CPU baseline is not handicapped by a naive triple loop.
typedef double v4d __attribute__((vector_size(32), aligned(8)));

// C (M x N, row-major, leading dimension ldc) += alpha * A (M x K, lda) * op(B), op(B) = B (K x N, ldb) or, with
// transB, B^T for B stored N x K.
void gemmCore(int M, int N, int K, const double* A, int lda, const double* B, int ldb, bool transB, double* C, int ldc,
    double alpha) {
    if (M == 0 || N == 0 || K == 0) return;
    constexpr int KB = 256, NR = 8, MR = 6;
    const int NP = (N + NR - 1) / NR;  // number of 8-wide column panels
    std::vector<double> Bp(size_t(NP) * KB * NR);
    for (int k0 = 0; k0 < K; k0 += KB) {
        const int kb = std::min(KB, K - k0);
        // pack op(B)[k0:k0+kb, :] into panels [panel][k][8], zero padded
        for (int pn = 0; pn < NP; ++pn) {
            double* dst = &Bp[size_t(pn) * KB * NR];
            for (int k = 0; k < kb; ++k)
                for (int j = 0; j < NR; ++j) {
                    const int col = pn * NR + j;
                    double v = 0.0;
                    if (col < N) v = transB ? B[size_t(col) * ldb + k0 + k] : B[size_t(k0 + k) * ldb + col];
                    dst[k * NR + j] = v;
                }
        }
        for (int i0 = 0; i0 < M; i0 += MR) {
            const int mr = std::min(MR, M - i0);
            const double* a[MR];
            for (int r = 0; r < MR; ++r) a[r] = &A[size_t(std::min(i0 + r, M - 1)) * lda + k0];
            for (int pn = 0; pn < NP; ++pn) {
                const double* b = &Bp[size_t(pn) * KB * NR];
                v4d c[MR][2];
                for (int r = 0; r < MR; ++r) c[r][0] = c[r][1] = v4d{0, 0, 0, 0};
                for (int k = 0; k < kb; ++k) {
                    const v4d b0 = *reinterpret_cast<const v4d*>(b + k * NR);
                    const v4d b1 = *reinterpret_cast<const v4d*>(b + k * NR + 4);
                    for (int r = 0; r < MR; ++r) {
                        const double ar = a[r][k];
                        const v4d av = {ar, ar, ar, ar};
                        c[r][0] += av * b0;
                        c[r][1] += av * b1;
                    }
                }
                const int nc = std::min(NR, N - pn * NR);
                for (int r = 0; r < mr; ++r) {
                    double* crow = &C[size_t(i0 + r) * ldc + pn * NR];
                    for (int j = 0; j < nc; ++j) crow[j] += alpha * (j < 4 ? c[r][0][j] : c[r][1][j - 4]);
                }
            }
        }
    }
}

Mat gemm(const Mat& A, const Mat& Bin, bool transB = false) {
    const int M = A.r, K = A.c, N = transB ? Bin.r : Bin.c;
    Mat C(M, N);
    gemmCore(M, N, K, A.d.data(), A.c, Bin.d.data(), Bin.c, transB, C.d.data(), N, 1.0);
    return C;
}
}  // namespace

// NB: like Eigen's dense product, gemm() does not skip structural zeros: the timed CPU baseline
// executes the reference's dense flop count.

namespace {
// General inverse by LU with partial pivoting (Eigen MatrixXd::inverse() -> PartialPivLU).
Mat inverseLU(const Mat& Ain) {
    const int n = Ain.r;
    Mat A = Ain;
    std::vector<int> piv(n);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::abs(A(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::abs(A(i, k)) > best) {
                best = std::abs(A(i, k));
                p = i;
            }
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(A(k, j), A(p, j));
            std::swap(piv[k], piv[p]);
        }
        const double inv = 1.0 / A(k, k);
        for (int i = k + 1; i < n; ++i) {
            const double l = A(i, k) * inv;
            A(i, k) = l;
            if (l == 0) continue;
            double* ai = &A.d[size_t(i) * n];
            const double* ak = &A.d[size_t(k) * n];
            for (int j = k + 1; j < n; ++j) ai[j] -= l * ak[j];
        }
    }
    // Solve A X = P I, row-oriented so the inner loops vectorise.
    Mat X(n, n);
    for (int i = 0; i < n; ++i) X(i, piv[i]) = 1.0;  // row i of P*I
    for (int i = 0; i < n; ++i) {                    // forward: L y = b
        double* xi = &X.d[size_t(i) * n];
        for (int k = 0; k < i; ++k) {
            const double l = A(i, k);
            if (l == 0) continue;
            const double* xk = &X.d[size_t(k) * n];
            for (int j = 0; j < n; ++j) xi[j] -= l * xk[j];
        }
    }
    for (int i = n - 1; i >= 0; --i) {  // backward: U x = y
        double* xi = &X.d[size_t(i) * n];
        for (int k = i + 1; k < n; ++k) {
            const double u = A(i, k);
            if (u == 0) continue;
            const double* xk = &X.d[size_t(k) * n];
            for (int j = 0; j < n; ++j) xi[j] -= u * xk[j];
        }
        const double inv = 1.0 / A(i, i);
        for (int j = 0; j < n; ++j) xi[j] *= inv;
    }
    return X;
}

// Householder QR solve of a small square system (Eigen householderQr().solve()).
void qrSolve(int n, std::vector<double> A, std::vector<double> b, double* x) {
    for (int k = 0; k < n; ++k) {
        double nrm = 0;
        for (int i = k; i < n; ++i) nrm += A[i * n + k] * A[i * n + k];
        nrm = std::sqrt(nrm);
        if (nrm == 0) continue;
        const double alpha = A[k * n + k] > 0 ? -nrm : nrm;
        std::vector<double> v(n, 0.0);
        for (int i = k; i < n; ++i) v[i] = A[i * n + k];
        v[k] -= alpha;
        double vv = 0;
        for (int i = k; i < n; ++i) vv += v[i] * v[i];
        if (vv == 0) continue;
        for (int j = k; j < n; ++j) {
            double s = 0;
            for (int i = k; i < n; ++i) s += v[i] * A[i * n + j];
            s = 2 * s / vv;
            for (int i = k; i < n; ++i) A[i * n + j] -= s * v[i];
        }
        double s = 0;
        for (int i = k; i < n; ++i) s += v[i] * b[i];
        s = 2 * s / vv;
        for (int i = k; i < n; ++i) b[i] -= s * v[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * x[j];
        x[i] = s / A[i * n + i];
    }
}

// ------------------------------------------------------------------ structured backend ("cpu_structured")
// The same filter equations evaluated without the reference's structural-zero work: F = I + T*A_b and C_b as sparse row
// lists, S^-1 / Sigma_e^-1 through Cholesky factors and forward substitutions instead of explicit inverses.  Exact-
// arithmetic identities only (K*delta = Y^T L^-1 delta, K C Sigma = Y^T Y with S = L L^T, Y = L^-1 C Sigma); it differs
// from the dense evaluation by rounding.  tests/test_oracle_structured.py holds it against the dense backend; it exists
// so that N >= 1000 parity tests are affordable and so that bench.py can report the algorithmic share of the speed-up.
struct SparseRows {  // row i: entries [start[i], start[i+1]) of (col, val)
    int rows = 0;
    std::vector<int> start, col;
    std::vector<double> val;
    SparseRows() : start(1, 0) {}
    void push(int c, double v) {
        col.push_back(c);
        val.push_back(v);
    }
    void endRow() {
        start.push_back(int(col.size()));
        ++rows;
    }
};
// out (A.rows x B.c) = A * B for dense row-major B; `out` is reshaped and overwritten (caller-owned scratch: no allocation
// once it has the size).
void sparseTimesDense(const SparseRows& A, const Mat& B, Mat& out) {
    out.r = A.rows;
    out.c = B.c;
    out.d.resize(size_t(A.rows) * B.c);
    for (int i = 0; i < A.rows; ++i) {
        double* o = &out.d[size_t(i) * B.c];
        std::fill(o, o + B.c, 0.0);
        for (int e = A.start[i]; e < A.start[i + 1]; ++e) {
            const double v = A.val[e];
            const double* b = &B.d[size_t(A.col[e]) * B.c];
            for (int j = 0; j < B.c; ++j) o[j] += v * b[j];
        }
    }
}
void transposeInto(const Mat& A, Mat& T) {
    T.r = A.c;
    T.c = A.r;
    T.d.resize(A.d.size());
    constexpr int TB = 16;
    for (int i0 = 0; i0 < A.r; i0 += TB)
        for (int j0 = 0; j0 < A.c; j0 += TB) {
            const int i1 = std::min(A.r, i0 + TB), j1 = std::min(A.c, j0 + TB);
            for (int j = j0; j < j1; ++j)
                for (int i = i0; i < i1; ++i) T.d[size_t(j) * A.r + i] = A.d[size_t(i) * A.c + j];
        }
}
// C (n x n, ldc) -= A A^T for A (n x k, lda): lower block triangle by gemmCore, then mirrored (the result is symmetric).
void syrkLowerMinus(int n, int k, const double* A, int lda, double* C, int ldc) {
    constexpr int RB = 96;
    for (int i0 = 0; i0 < n; i0 += RB) {
        const int ib = std::min(RB, n - i0);
        gemmCore(ib, i0 + ib, k, &A[size_t(i0) * lda], lda, A, lda, true, &C[size_t(i0) * ldc], ldc, -1.0);
    }
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) C[size_t(i) * ldc + j] = C[size_t(j) * ldc + i];
}
// In-place lower Cholesky A = L L^T of the leading n x n block (row-major, leading dimension ld), blocked right-looking;
// the strict upper triangle is neither read nor kept meaningful.  Returns false if a pivot <= 0.
bool choleskyBlocked(double* A, int n, int ld) {
    constexpr int NB = 64;
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = std::min(NB, n - k0);
        for (int j = k0; j < k0 + kb; ++j) {  // unblocked factorisation of the diagonal block
            double d = A[size_t(j) * ld + j];
            for (int k = k0; k < j; ++k) d -= A[size_t(j) * ld + k] * A[size_t(j) * ld + k];
            if (!(d > 0)) return false;
            d = std::sqrt(d);
            A[size_t(j) * ld + j] = d;
            for (int i = j + 1; i < k0 + kb; ++i) {
                double v = A[size_t(i) * ld + j];
                for (int k = k0; k < j; ++k) v -= A[size_t(i) * ld + k] * A[size_t(j) * ld + k];
                A[size_t(i) * ld + j] = v / d;
            }
        }
        const int r0 = k0 + kb, rest = n - r0;
        if (rest <= 0) break;
        for (int i = r0; i < n; ++i) {  // panel: L_ik = A_ik L_kk^-T, row by row
            double* a = &A[size_t(i) * ld + k0];
            for (int j = 0; j < kb; ++j) {
                const double* l = &A[size_t(k0 + j) * ld + k0];
                double v = a[j];
                for (int k = 0; k < j; ++k) v -= a[k] * l[k];
                a[j] = v / l[j];
            }
        }
        // trailing block -= panel * panel^T, lower block triangle only (the upper half is never read)
        constexpr int RB = 96;
        for (int i0 = 0; i0 < rest; i0 += RB) {
            const int ib = std::min(RB, rest - i0);
            gemmCore(ib, i0 + ib, kb, &A[size_t(r0 + i0) * ld + k0], ld, &A[size_t(r0) * ld + k0], ld, true,
                &A[size_t(r0 + i0) * ld + r0], ld, -1.0);
        }
    }
    return true;
}
// W (n x r, leading dimension ldw) <- L^-1 W for the lower factor L (n x n, ld), blocked forward substitution.
void forwardSolve(const double* L, int n, int ld, double* W, int r, int ldw) {
    constexpr int NB = 64;
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = std::min(NB, n - k0);
        if (k0 > 0) gemmCore(kb, r, k0, &L[size_t(k0) * ld], ld, W, ldw, false, &W[size_t(k0) * ldw], ldw, -1.0);
        for (int i = k0; i < k0 + kb; ++i) {
            double* wi = &W[size_t(i) * ldw];
            for (int k = k0; k < i; ++k) {
                const double l = L[size_t(i) * ld + k];
                const double* wk = &W[size_t(k) * ldw];
                for (int j = 0; j < r; ++j) wi[j] -= l * wk[j];
            }
            const double inv = 1.0 / L[size_t(i) * ld + i];
            for (int j = 0; j < r; ++j) wi[j] *= inv;
        }
    }
}

// ------------------------------------------------------------------ state / group types
struct IMU {
    double stamp = 0;
    V3 omega, accel;
};
inline IMU imuScale(const IMU& a, double c) { return {a.stamp, c * a.omega, c * a.accel}; }                 // IMUVelocity.cpp:53-58
inline IMU imuAdd(const IMU& a, const IMU& b) { return {a.stamp > 0 ? a.stamp : b.stamp, a.omega + b.omega, a.accel + b.accel}; }  // :35-41

struct State {  // VIOState.h:51-60
    SE3 pose;
    V3 velocity;
    std::vector<V3> p;
    std::vector<int> id;
    SE3 cameraOffset;
};
struct Manifold {  // VIOState.h:43-49
    V3 gravityDir, velocity;
    std::vector<V3> p;
    std::vector<int> id;
    SE3 cameraOffset;
};
struct Group {  // VIOGroup.h:24-33
    SE3 A;
    V3 w;
    std::vector<SOT3> Q;
    std::vector<int> id;
};

Manifold projectToManifold(const State& Xi) {  // VIOState.cpp:88-95
    return {rotate(inverse(Xi.pose.q), V3{0, 0, 1}), Xi.velocity, Xi.p, Xi.id, Xi.cameraOffset};
}
State stateGroupAction(const Group& X, const State& s) {  // VIOGroup.cpp:23-45
    State n;
    n.pose = s.pose * X.A;
    n.velocity = rotate(inverse(X.A.q), s.velocity - X.w);
    n.cameraOffset = s.cameraOffset;
    n.id = s.id;
    n.p.resize(s.p.size());
    for (size_t i = 0; i < s.p.size(); ++i) n.p[i] = X.Q[i].inv().apply(s.p[i]);
    return n;
}
Manifold stateGroupAction(const Group& X, const Manifold& s) {  // VIOGroup.cpp:47-69
    Manifold n;
    n.gravityDir = rotate(inverse(X.A.q), s.gravityDir);
    n.velocity = rotate(inverse(X.A.q), s.velocity - X.w);
    n.cameraOffset = s.cameraOffset;
    n.id = s.id;
    n.p.resize(s.p.size());
    for (size_t i = 0; i < s.p.size(); ++i) n.p[i] = X.Q[i].inv().apply(s.p[i]);
    return n;
}
Group groupMul(const Group& a, const Group& b) {  // VIOGroup.cpp:92-110
    Group r;
    r.A = a.A * b.A;
    r.w = a.w + rotate(a.A.q, b.w);
    r.Q.resize(a.Q.size());
    for (size_t i = 0; i < a.Q.size(); ++i) r.Q[i] = a.Q[i] * b.Q[i];
    r.id = a.id;
    return r;
}
Group groupInverse(const Group& a) {  // VIOGroup.cpp:124-134
    Group r;
    r.A = a.A.inv();
    r.w = -rotate(inverse(a.A.q), a.w);
    r.Q.resize(a.Q.size());
    for (size_t i = 0; i < a.Q.size(); ++i) r.Q[i] = a.Q[i].inv();
    r.id = a.id;
    return r;
}

Group liftVelocityDiscrete(const Manifold& st, const IMU& vel, double dt) {  // VIOGroup.cpp:209-243
    Group lift;
    double AVel[6] = {vel.omega.x, vel.omega.y, vel.omega.z, st.velocity.x, st.velocity.y, st.velocity.z};
    double AVdt[6];
    for (int i = 0; i < 6; ++i) AVdt[i] = dt * AVel[i];
    lift.A = SE3Exp(AVdt);
    lift.w = st.velocity - rotate(lift.A.q, st.velocity + dt * (-(skew(vel.omega) * st.velocity) + vel.accel -
                                                                     GRAVITY_CONSTANT * st.gravityDir));
    double U_C[6], mU[6];
    adjointApply(st.cameraOffset.inv(), AVel, U_C);
    for (int i = 0; i < 6; ++i) mU[i] = -dt * U_C[i];
    const SE3 camInv = SE3Exp(mU);
    const size_t N = st.p.size();
    lift.Q.resize(N);
    lift.id = st.id;
    for (size_t i = 0; i < N; ++i) {
        const V3 p0 = st.p[i];
        const V3 p1 = camInv.apply(p0);
        lift.Q[i].q = SO3FromVectors(normalized(p1), normalized(p0));
        lift.Q[i].a = norm(p0) / norm(p1);
    }
    return lift;
}
// liftVelocity + VIOExp (VIOGroup.cpp:178-207, :245-255) composed: X_lift = VIOExp(dt * Lambda)
Group liftVelocityExp(const Manifold& st, const IMU& vel, double dt) {
    Group r;
    double U[6] = {vel.omega.x, vel.omega.y, vel.omega.z, st.velocity.x, st.velocity.y, st.velocity.z};
    double Udt[6];
    for (int i = 0; i < 6; ++i) Udt[i] = dt * U[i];
    r.A = SE3Exp(Udt);
    r.w = dt * (-vel.accel + GRAVITY_CONSTANT * st.gravityDir);
    double U_C[6];
    adjointApply(st.cameraOffset.inv(), U, U_C);
    const V3 omega_C{U_C[0], U_C[1], U_C[2]}, v_C{U_C[3], U_C[4], U_C[5]};
    r.id = st.id;
    r.Q.resize(st.p.size());
    for (size_t i = 0; i < st.p.size(); ++i) {
        const V3 p = st.p[i];
        const double n2 = dot(p, p);
        const V3 Wr = omega_C + (1.0 / n2) * (skew(p) * v_C);
        const double Ws = dot(p, v_C) / n2;
        r.Q[i].q = SO3Exp(dt * Wr);      // SOT3.cpp:127-132
        r.Q[i].a = std::exp(dt * Ws);
    }
    return r;
}

// ------------------------------------------------------------------ EqF matrices (EqFMatrices.cpp)
Mat EqFStateMatrixA(const Group& X, const Manifold& xi0, const IMU& imuVel) {  // :277-317
    const int N = int(xi0.p.size());
    Mat A0t(5 + 3 * N, 5 + 3 * N);
    const double zero2[2] = {0, 0};
    const M32 ID = stereoSphereChartInvDiff(zero2, xi0.gravityDir);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) A0t(2 + i, j) = -ID.m[i][j] * GRAVITY_CONSTANT;
    const M3 R_IC = toMatrix(xi0.cameraOffset.q);
    const M3 R_Ahat = toMatrix(X.A.q);
    for (int i = 0; i < N; ++i) {
        const M3 Qhat = X.Q[i].a * toMatrix(X.Q[i].q);
        setBlock(A0t, 5 + 3 * i, 2, Qhat * transpose(R_IC) * transpose(R_Ahat), -1.0);
    }
    const Manifold xi_hat = stateGroupAction(X, xi0);
    const double U_I[6] = {imuVel.omega.x, imuVel.omega.y, imuVel.omega.z, xi_hat.velocity.x, xi_hat.velocity.y, xi_hat.velocity.z};
    double U_C[6];
    adjointApply(xi0.cameraOffset.inv(), U_I, U_C);
    const V3 v_C{U_C[3], U_C[4], U_C[5]};
    for (int i = 0; i < N; ++i) {
        const M3 Qhat = X.Q[i].a * toMatrix(X.Q[i].q);
        const V3 qhat = xi_hat.p[i];
        const M3 inner = skew(qhat) * skew(v_C) - 2.0 * outer(v_C, qhat) + outer(qhat, v_C);
        const M3 A_qi = (1 / dot(qhat, qhat)) * (Qhat * inner * inverse3(Qhat));
        setBlock(A0t, 5 + 3 * i, 5 + 3 * i, A_qi, -1.0);
    }
    return A0t;
}
Mat EqFOutputMatrixC(const Manifold& xi0) {  // :319-344
    const int N = int(xi0.p.size());
    Mat C0(2 * N, 5 + 3 * N);
    for (int i = 0; i < N; ++i) {
        const V3 qi0 = xi0.p[i];
        const V3 yi0 = normalized(qi0);
        const M23 D = stereoSphereChartDiff(yi0, yi0);
        const M3 P = M3::I() - outer(yi0, yi0);
        const double s = 1 / norm(qi0);
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) {
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += (s * D.m[r][k]) * P.m[k][c];
                C0(2 * i + r, 5 + 3 * i + c) = acc;
            }
    }
    return C0;
}
Mat EqFInputMatrixB(const Group& X, const Manifold& xi0) {  // :346-382
    const int N = int(xi0.p.size());
    Mat Bt(5 + 3 * N, 6);
    const Manifold xi_hat = stateGroupAction(X, xi0);
    const M3 R_A = toMatrix(X.A.q);
    const M23 D = stereoSphereChartDiff(xi0.gravityDir, xi0.gravityDir);
    const M3 RAg = R_A * skew(xi_hat.gravityDir);
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += D.m[r][k] * RAg.m[k][c];
            Bt(r, c) = acc;
        }
    setBlock(Bt, 2, 0, R_A * skew(xi_hat.velocity));
    setBlock(Bt, 2, 3, R_A);
    const M3 RT_IC = toMatrix(inverse(xi0.cameraOffset.q));
    const V3 x_IC = xi0.cameraOffset.x;
    for (int i = 0; i < N; ++i) {
        const M3 Qhat = X.Q[i].a * toMatrix(X.Q[i].q);
        setBlock(Bt, 5 + 3 * i, 0, Qhat * (skew(xi_hat.p[i]) * RT_IC + RT_IC * skew(x_IC)));
    }
    return Bt;
}

std::vector<double> bundleLift(const std::vector<double>& base, const State& xi0, const Group& X, const Mat& Sigma,
    bool structured = false) {  // :173-252
    const State xiHat = stateGroupAction(X, xi0);
    const int N = int(xi0.p.size());
    const V3 eta0 = normalized(projectToManifold(xi0).gravityDir);
    const double zero2[2] = {0, 0};
    const M32 ID = stereoSphereChartInvDiff(zero2, eta0);
    V3 t{ID.m[0][0] * base[0] + ID.m[0][1] * base[1], ID.m[1][0] * base[0] + ID.m[1][1] * base[1],
        ID.m[2][0] * base[0] + ID.m[2][1] * base[1]};
    const V3 dUw = -(skew(eta0) * t);
    double DeltaU[6] = {dUw.x, dUw.y, dUw.z, 0, 0, 0};
    // KPara (6x4), KPerp (6x6) -- :186-199 (the :200 assignment writes zeros into zeros)
    double KPara[6][4] = {};
    KPara[0][0] = eta0.x; KPara[1][0] = eta0.y; KPara[2][0] = eta0.z;
    KPara[3][1] = KPara[4][2] = KPara[5][3] = 1;
    const M3 Pperp = M3::I() - outer(eta0, eta0);
    const V3 fixedW = Pperp * V3{DeltaU[0], DeltaU[1], DeltaU[2]};
    const double DeltaUFixed[6] = {fixedW.x, fixedW.y, fixedW.z, 0, 0, 0};
    const Quat R_C = xiHat.pose.q * xiHat.cameraOffset.q;
    const M3 R_CT = toMatrix(inverse(R_C));
    // AdP0 (6x6)
    double AdP0[6][6] = {};
    {
        const M3 R = toMatrix(xi0.pose.q);
        const M3 xR = skew(xi0.pose.x) * R;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                AdP0[i][j] = R.m[i][j];
                AdP0[3 + i][j] = xR.m[i][j];
                AdP0[3 + i][3 + j] = R.m[i][j];
            }
    }
    Mat coeff(3 * N, 4), D(5 + 3 * N, 3 * N);
    Mat obs(3 * N, 1);
    const SE3 PC = xiHat.pose * xiHat.cameraOffset;
    for (int i = 0; i < N; ++i) {
        const V3 g{base[5 + 3 * i], base[6 + 3 * i], base[7 + 3 * i]};
        const V3 pHat = PC.apply(xiHat.p[i]);
        const V3 alpha = -rotate(R_C, X.Q[i].inv().apply(g));
        double pHatMat[3][6] = {};
        const M3 ms = skew(pHat);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) pHatMat[r][c] = -ms.m[r][c];
            pHatMat[r][3 + r] = 1;
        }
        double PA[3][6];  // pHatMat * AdP0
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 6; ++c) {
                double s = 0;
                for (int k = 0; k < 6; ++k) s += pHatMat[r][k] * AdP0[k][c];
                PA[r][c] = s;
            }
        for (int r = 0; r < 3; ++r) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += PA[r][k] * DeltaUFixed[k];
            obs(3 * i + r, 0) = alpha[r] - s;
            for (int c = 0; c < 4; ++c) {
                double q = 0;
                for (int k = 0; k < 6; ++k) q += PA[r][k] * KPara[k][c];
                coeff(3 * i + r, c) = q;
            }
        }
        setBlock(D, 5 + 3 * i, 3 * i, X.Q[i].asMatrix3() * R_CT);
    }
    Mat lhs(4, 4), rhs(4, 1);
    bool done = false;
    if (structured) {
        // coeff^T (D^T Sigma^-1 D) [coeff | obs] = (L^-1 D coeff)^T (L^-1 D [coeff | obs]) with Sigma = L L^T
        const int ne = 5 + 3 * N;
        Mat Z(ne, 5);
        for (int i = 0; i < N; ++i)
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 5; ++c) {
                    double acc = 0;
                    for (int k = 0; k < 3; ++k) acc += D(5 + 3 * i + r, 3 * i + k) * (c < 4 ? coeff(3 * i + k, c) : obs(3 * i + k, 0));
                    Z(5 + 3 * i + r, c) = acc;
                }
        Mat L = Sigma;
        if (choleskyBlocked(L.d.data(), ne, ne)) {
            forwardSolve(L.d.data(), ne, ne, Z.d.data(), 5, 5);
            for (int a = 0; a < 4; ++a) {
                for (int b = 0; b < 4; ++b) {
                    double acc = 0;
                    for (int i = 0; i < ne; ++i) acc += Z(i, a) * Z(i, b);
                    lhs(a, b) = acc;
                }
                double acc = 0;
                for (int i = 0; i < ne; ++i) acc += Z(i, a) * Z(i, 4);
                rhs(a, 0) = acc;
            }
            done = true;
        }
    }
    if (!done) {
        // weightMat = D^T * Sigma^-1 * D   (:239, explicit dense inverse, evaluated left to right)
        Mat Dt(D.c, D.r);
        for (int i = 0; i < D.r; ++i)
            for (int j = 0; j < D.c; ++j) Dt(j, i) = D(i, j);
        const Mat W = gemm(gemm(Dt, inverseLU(Sigma)), D);
        // (coeff^T W coeff) x = coeff^T W obs
        Mat coeffT(4, 3 * N);
        for (int i = 0; i < 3 * N; ++i)
            for (int j = 0; j < 4; ++j) coeffT(j, i) = coeff(i, j);
        const Mat cTW = gemm(coeffT, W);
        lhs = gemm(cTW, coeff);
        rhs = gemm(cTW, obs);
    }
    double sol[4];
    qrSolve(4, lhs.d, rhs.d, sol);
    std::vector<double> lifted(9 + 3 * N);
    for (int i = 0; i < 6; ++i) {
        double s = DeltaUFixed[i];
        for (int c = 0; c < 4; ++c) s += KPara[i][c] * sol[c];
        lifted[i] = s;
    }
    for (int i = 0; i < 3 + 3 * N; ++i) lifted[6 + i] = base[2 + i];
    return lifted;
}

Group liftTotalSpaceInnovationDiscrete(const std::vector<double>& G, const State& xi0) {  // :254-275
    Group lift;
    lift.A = SE3Exp(G.data());
    lift.w = xi0.velocity - rotate(lift.A.q, xi0.velocity + V3{G[6], G[7], G[8]});
    const size_t N = xi0.p.size();
    lift.id = xi0.id;
    lift.Q.resize(N);
    for (size_t i = 0; i < N; ++i) {
        const V3 qi = xi0.p[i];
        const V3 qi1 = qi + V3{G[9 + 3 * i], G[10 + 3 * i], G[11 + 3 * i]};
        lift.Q[i].q = SO3FromVectors(normalized(qi1), normalized(qi));
        lift.Q[i].a = norm(qi) / norm(qi1);
    }
    return lift;
}
// VIOExp(liftTotalSpaceInnovation(G, xi0)) (EqFMatrices.cpp:69-96) / VIOExp(liftInnovation(g, xi0)) (:35-67)
Group liftAlgebraExp(const double U[6], V3 u, const std::vector<double>& lm, int off, const State& xi0) {
    Group r;
    r.A = SE3Exp(U);
    r.w = u;
    r.id = xi0.id;
    r.Q.resize(xi0.p.size());
    for (size_t i = 0; i < xi0.p.size(); ++i) {
        const V3 g{lm[off + 3 * i], lm[off + 3 * i + 1], lm[off + 3 * i + 2]};
        const V3 q = xi0.p[i];
        const double n2 = dot(q, q);
        r.Q[i].q = SO3Exp((-1.0 / n2) * cross(q, g));
        r.Q[i].a = std::exp(-dot(q, g) / n2);
    }
    return r;
}

// ------------------------------------------------------------------ settings + filter
struct Settings {  // VIOFilterSettings.h:28-54
    double biasOmegaProcessVariance = 0.001, biasAccelProcessVariance = 0.001, gravityProcessVariance = 0.001,
           velocityProcessVariance = 0.001, pointProcessVariance = 0.001, velOmegaVariance = 0.1, velAccelVariance = 0.1,
           measurementVariance = 0.1, initialGravityVariance = 1.0, initialVelocityVariance = 1.0,
           initialPointVariance = 1.0, initialBiasOmegaVariance = 1.0, initialBiasAccelVariance = 1.0,
           initialSceneDepth = 1.0, outlierThreshold = 0.01;
    int useInnovationLift = 1, useDiscreteInnovationLift = 1, useDiscreteVelocityLift = 1, fastRiccati = 0;
    double initialAccelBias[3] = {0, 0, 0}, initialOmegaBias[3] = {0, 0, 0};
    double cameraOffset_x[3] = {0, 0, 0};
    double cameraOffset_q[4] = {1, 0, 0, 0};
};

struct Filter {
    Settings s;
    double inputBias[6] = {0, 0, 0, 0, 0, 0};
    State xi0;
    Group X;
    Mat Sigma;
    bool initialised = false;
    double currentTime = -1;
    IMU currentVelocity, accumulatedVelocity;
    double accumulatedTime = 0;
    // internals of the last update (for kernel-level parity tests)
    std::vector<double> lastDelta, lastGamma, lastGammaTotal;
    Mat lastS;
    bool structured = false;  // cpu_structured backend (same equations, no structural-zero work, Cholesky-form update)
    Mat scrFS, scrFSt, scrG;  // scratch of riccatiStructured (kept between calls)

    explicit Filter(const Settings& st) : s(st) {  // VIOFilter.cpp:60-73
        Sigma = Mat::Identity(SIGMA_BASE_SIZE);
        for (int i = 0; i < 3; ++i) {
            Sigma(i, i) = s.initialBiasOmegaVariance;
            Sigma(3 + i, 3 + i) = s.initialBiasAccelVariance;
            Sigma(8 + i, 8 + i) = s.initialVelocityVariance;
        }
        Sigma(6, 6) = Sigma(7, 7) = s.initialGravityVariance;
        xi0.cameraOffset.q = {s.cameraOffset_q[0], s.cameraOffset_q[1], s.cameraOffset_q[2], s.cameraOffset_q[3]};
        xi0.cameraOffset.x = {s.cameraOffset_x[0], s.cameraOffset_x[1], s.cameraOffset_x[2]};
        for (int i = 0; i < 3; ++i) {
            inputBias[i] = s.initialOmegaBias[i];
            inputBias[3 + i] = s.initialAccelBias[i];
        }
    }
    State stateEstimate() const { return stateGroupAction(X, xi0); }  // :304

    void processIMUData(const IMU& imu) {  // :120-131
        IMU unbiased = imu;
        unbiased.omega = imu.omega - V3{inputBias[0], inputBias[1], inputBias[2]};
        unbiased.accel = imu.accel - V3{inputBias[3], inputBias[4], inputBias[5]};
        if (!initialised) {  // :133-144
            xi0.pose = SE3{};
            xi0.velocity = V3{};
            initialised = true;
            xi0.pose.q = SO3FromVectors(normalized(unbiased.accel), V3{0, 0, 1});
        }
        integrateUpToTime(imu.stamp, !s.fastRiccati);
        currentVelocity = unbiased;
        currentTime = imu.stamp;
    }

    bool integrateUpToTime(double newTime, bool doRiccati) {  // :146-209
        if (currentTime < 0) return false;
        const double dt = newTime - currentTime;
        if (dt <= 0) return false;
        accumulatedTime += dt;
        accumulatedVelocity = imuAdd(accumulatedVelocity, imuScale(currentVelocity, dt));
        const int N = int(xi0.p.size());
        const State currentState = stateEstimate();
        if (doRiccati && structured) {
            riccatiStructured(N);
            accumulatedVelocity = IMU{};
            accumulatedTime = 0.0;
        } else if (doRiccati) {
            const int n = Sigma.r;
            Mat PMat = Mat::Identity(n);
            for (int i = 0; i < 3; ++i) {
                PMat(i, i) *= s.biasOmegaProcessVariance;
                PMat(3 + i, 3 + i) *= s.biasAccelProcessVariance;
                PMat(8 + i, 8 + i) *= s.velocityProcessVariance;
            }
            PMat(6, 6) *= s.gravityProcessVariance;
            PMat(7, 7) *= s.gravityProcessVariance;
            for (int i = 0; i < 3 * N; ++i) PMat(11 + i, 11 + i) *= s.pointProcessVariance;
            const Manifold xi0m = projectToManifold(xi0);
            const Mat A0t = EqFStateMatrixA(X, xi0m, imuScale(accumulatedVelocity, 1.0 / accumulatedTime));
            const Mat Bt = EqFInputMatrixB(X, xi0m);
            Mat R = Mat::Identity(6);
            for (int i = 0; i < 3; ++i) {
                R(i, i) *= s.velOmegaVariance;
                R(3 + i, 3 + i) *= s.velAccelVariance;
            }
            Mat Ab(n, n);
            for (int i = 0; i < A0t.r; ++i) {
                for (int j = 0; j < A0t.c; ++j) Ab(6 + i, 6 + j) = A0t(i, j);
                for (int j = 0; j < 6; ++j) Ab(6 + i, j) = -Bt(i, j);
            }
            Mat F = Mat::Identity(n);
            for (size_t k = 0; k < F.d.size(); ++k) F.d[k] += Ab.d[k] * accumulatedTime;
            Mat Bb(n, 6);
            for (int i = 0; i < Bt.r; ++i)
                for (int j = 0; j < 6; ++j) Bb(6 + i, j) = Bt(i, j);
            const Mat BRBt = gemm(gemm(Bb, R), Bb, true);
            const Mat FSFt = gemm(gemm(F, Sigma), F, true);
            for (size_t k = 0; k < Sigma.d.size(); ++k)
                Sigma.d[k] = accumulatedTime * (PMat.d[k] + BRBt.d[k]) + FSFt.d[k];
            accumulatedVelocity = IMU{};
            accumulatedTime = 0.0;
        }
        const Manifold cur = projectToManifold(currentState);
        if (s.useDiscreteVelocityLift)
            X = groupMul(X, liftVelocityDiscrete(cur, currentVelocity, dt));
        else
            X = groupMul(X, liftVelocityExp(cur, currentVelocity, dt));
        currentTime = newTime;
        return true;
    }

    // VIOFilter.cpp:160-194 with F = I + T*A_b as sparse rows (9 non-zeros per landmark row): Sigma <- T(P + B R B^T) +
    // F Sigma F^T evaluated as (F (F Sigma)^T)^T, the rank-6 input noise term from the 6 columns of B.
    void riccatiStructured(int N) {
        const int n = Sigma.r;
        const double T = accumulatedTime;
        const Manifold xi0m = projectToManifold(xi0);
        const Mat A0t = EqFStateMatrixA(X, xi0m, imuScale(accumulatedVelocity, 1.0 / accumulatedTime));
        const Mat Bt = EqFInputMatrixB(X, xi0m);
        SparseRows F;
        for (int i = 0; i < 6; ++i) {
            F.push(i, 1.0);
            F.endRow();
        }
        for (int i = 0; i < A0t.r; ++i) {
            for (int j = 0; j < 6; ++j)
                if (Bt(i, j) != 0.0) F.push(j, -Bt(i, j) * T);
            for (int j = 0; j < A0t.c; ++j) {
                const double v = (i == j ? 1.0 : 0.0) + A0t(i, j) * T;
                if (v != 0.0) F.push(6 + j, v);
            }
            F.endRow();
        }
        sparseTimesDense(F, Sigma, scrFS);
        transposeInto(scrFS, scrFSt);  // = Sigma F^T (Sigma is symmetric to rounding; the reference multiplies F Sigma first too)
        sparseTimesDense(F, scrFSt, scrG);  // = (F Sigma) F^T transposed
        const Mat& G = scrG;
        const double Rw = s.velOmegaVariance, Ra = s.velAccelVariance;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double q = 0;
                if (i >= 6 && j >= 6) {
                    const double* bi = &Bt.d[size_t(i - 6) * 6];
                    const double* bj = &Bt.d[size_t(j - 6) * 6];
                    q = Rw * (bi[0] * bj[0] + bi[1] * bj[1] + bi[2] * bj[2]) + Ra * (bi[3] * bj[3] + bi[4] * bj[4] + bi[5] * bj[5]);
                }
                if (i == j) {
                    q += i < 3 ? s.biasOmegaProcessVariance
                       : i < 6 ? s.biasAccelProcessVariance
                       : i < 8 ? s.gravityProcessVariance
                       : i < 11 ? s.velocityProcessVariance
                                : s.pointProcessVariance;
                }
                Sigma(i, j) = T * q + G(j, i);
            }
        (void)N;
    }

    void removeLandmarkAtIndex(int idx) {  // :421-427
        xi0.p.erase(xi0.p.begin() + idx);
        xi0.id.erase(xi0.id.begin() + idx);
        X.id.erase(X.id.begin() + idx);
        X.Q.erase(X.Q.begin() + idx);
        const int r0 = SIGMA_BASE_SIZE + 3 * idx, n = Sigma.r;
        Mat Sn(n - 3, n - 3);
        for (int i = 0, ii = 0; i < n; ++i) {
            if (i >= r0 && i < r0 + 3) continue;
            for (int j = 0, jj = 0; j < n; ++j) {
                if (j >= r0 && j < r0 + 3) continue;
                Sn(ii, jj++) = Sigma(i, j);
            }
            ++ii;
        }
        Sigma = Sn;
    }

    void processVisionData(double stamp, int nb, const int* ids, const double* y) {  // :232-302
        lastDelta.clear();
        lastGamma.clear();
        lastGammaTotal.clear();
        if (!integrateUpToTime(stamp, true) || !initialised) return;
        // removeOldLandmarks :393-419
        for (int li = int(X.id.size()) - 1; li >= 0; --li)
            if (std::find(ids, ids + nb, X.id[li]) == ids + nb) removeLandmarkAtIndex(li);
        // matchMeasurementsToState :211-230
        std::vector<int> mid(nb);
        std::vector<V3> my(nb);
        {
            int newPos = int(X.id.size()) - 1;
            for (int k = 0; k < nb; ++k) {
                const auto it = std::find(X.id.begin(), X.id.end(), ids[k]);
                const int idx = (it != X.id.end()) ? int(it - X.id.begin()) : ++newPos;
                mid[idx] = ids[k];
                my[idx] = V3{y[3 * k], y[3 * k + 1], y[3 * k + 2]};
            }
        }
        // removeOutliers :429-443
        {
            const State xiHat = stateEstimate();
            for (int i = int(xiHat.p.size()) - 1; i >= 0; --i) {
                const double err = norm(my[i] - normalized(xiHat.p[i]));
                if (err > s.outlierThreshold) {
                    removeLandmarkAtIndex(i);
                    mid.erase(mid.begin() + i);
                    my.erase(my.begin() + i);
                }
            }
        }
        // addNewLandmarks :345-391
        {
            std::vector<int> fresh;
            for (size_t k = 0; k < mid.size(); ++k)
                if (std::find(X.id.begin(), X.id.end(), mid[k]) == X.id.end()) fresh.push_back(int(k));
            if (!fresh.empty()) {
                const State est = stateEstimate();
                std::vector<double> d2(est.p.size());
                for (size_t i = 0; i < est.p.size(); ++i) d2[i] = dot(est.p[i], est.p[i]);
                double medianDepth = s.initialSceneDepth;
                if (!d2.empty()) {
                    auto mid_it = d2.begin() + d2.size() / 2;
                    std::nth_element(d2.begin(), mid_it, d2.end());
                    medianDepth = std::pow(*mid_it, 0.5);
                }
                const int og = Sigma.r, k3 = 3 * int(fresh.size());
                for (int k : fresh) {
                    xi0.p.push_back(medianDepth * my[k]);
                    xi0.id.push_back(mid[k]);
                    X.id.push_back(mid[k]);
                    X.Q.push_back(SOT3{});
                }
                Mat Sn(og + k3, og + k3);
                for (int i = 0; i < og; ++i)
                    for (int j = 0; j < og; ++j) Sn(i, j) = Sigma(i, j);
                for (int i = 0; i < k3; ++i) Sn(og + i, og + i) = s.initialPointVariance;
                Sigma = Sn;
            }
        }
        if (mid.empty()) return;
        const int N = int(xi0.p.size());
        const int n = Sigma.r, m = 2 * N;
        // innovation :264-267
        std::vector<double> delta(m);
        const Group Xinv = groupInverse(X);
        for (int i = 0; i < N; ++i) {
            const V3 y0 = normalized(xi0.p[i]);                     // measureSystemState
            const V3 yerr = rotate(inverse(Xinv.Q[i].q), my[i]);    // outputGroupAction(X^-1, y)
            stereoSphereChart(yerr, y0, &delta[2 * i]);             // outputCoordinateChart
        }
        const Mat C0 = EqFOutputMatrixC(projectToManifold(xi0));
        Mat Cb(m, n);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < C0.c; ++j) Cb(i, 6 + j) = C0(i, j);
        Mat S, K, Yt;  // Yt (structured only): (L^-1 C Sigma)^T, n x m
        std::vector<double> gam(n, 0.0);
        bool chol = false;
        if (structured) {
            SparseRows Cs;
            for (int i = 0; i < m; ++i) {
                for (int j = 0; j < C0.c; ++j)
                    if (C0(i, j) != 0.0) Cs.push(6 + j, C0(i, j));
                Cs.endRow();
            }
            Mat CS;
            sparseTimesDense(Cs, Sigma, CS);
            S = Mat(m, m);
            for (int i = 0; i < m; ++i) {
                for (int j = 0; j < m; ++j) {
                    double acc = 0;
                    for (int e = Cs.start[j]; e < Cs.start[j + 1]; ++e) acc += CS(i, Cs.col[e]) * Cs.val[e];
                    S(i, j) = acc;
                }
                S(i, i) += s.measurementVariance;
            }
            Mat L = S;
            if (choleskyBlocked(L.d.data(), m, m)) {
                Mat W(m, n + 1);  // [C Sigma | delta]
                for (int i = 0; i < m; ++i) {
                    std::copy(&CS.d[size_t(i) * n], &CS.d[size_t(i) * n] + n, &W.d[size_t(i) * (n + 1)]);
                    W(i, n) = delta[i];
                }
                forwardSolve(L.d.data(), m, m, W.d.data(), n + 1, n + 1);
                for (int r = 0; r < m; ++r) {
                    const double z = W(r, n);
                    const double* y = &W.d[size_t(r) * (n + 1)];
                    for (int i = 0; i < n; ++i) gam[i] += y[i] * z;  // K delta = Y^T (L^-1 delta)
                }
                Yt = Mat(n, m);
                for (int r = 0; r < m; ++r)
                    for (int i = 0; i < n; ++i) Yt(i, r) = W(r, i);
                chol = true;
            }
        }
        if (!chol) {
            S = gemm(gemm(Cb, Sigma), Cb, true);
            for (int i = 0; i < m; ++i) S(i, i) += s.measurementVariance;
            K = gemm(gemm(Sigma, Cb, true), inverseLU(S));
            for (int i = 0; i < n; ++i) {
                double acc = 0;
                for (int j = 0; j < m; ++j) acc += K(i, j) * delta[j];
                gam[i] = acc;
            }
        }
        std::vector<double> gE(gam.begin() + 6, gam.end());
        Group Delta;
        if (s.useInnovationLift) {
            Mat Se(n - 6, n - 6);
            for (int i = 0; i < n - 6; ++i)
                for (int j = 0; j < n - 6; ++j) Se(i, j) = Sigma(6 + i, 6 + j);
            const std::vector<double> G = bundleLift(gE, xi0, X, Se, structured);
            lastGammaTotal = G;
            if (s.useDiscreteInnovationLift) {
                Delta = liftTotalSpaceInnovationDiscrete(G, xi0);
            } else {
                const V3 u = -V3{G[6], G[7], G[8]} - skew(V3{G[0], G[1], G[2]}) * xi0.velocity;
                Delta = liftAlgebraExp(G.data(), u, G, 9, xi0);
            }
        } else {  // liftInnovation(gE, xi0) :35-67
            const Manifold xm = projectToManifold(xi0);
            const double zero2[2] = {0, 0};
            const M32 ID = stereoSphereChartInvDiff(zero2, xm.gravityDir);
            const V3 t{ID.m[0][0] * gE[0] + ID.m[0][1] * gE[1], ID.m[1][0] * gE[0] + ID.m[1][1] * gE[1],
                ID.m[2][0] * gE[0] + ID.m[2][1] * gE[1]};
            const V3 Uw = -(skew(xm.gravityDir) * t);
            const double U[6] = {Uw.x, Uw.y, Uw.z, 0, 0, 0};
            const V3 u = -V3{gE[2], gE[3], gE[4]} - skew(Uw) * xm.velocity;
            Delta = liftAlgebraExp(U, u, gE, 5, xi0);
        }
        lastDelta = delta;
        lastGamma = gam;
        lastS = S;
        for (int i = 0; i < 6; ++i) inputBias[i] += gam[i];
        X = groupMul(Delta, X);
        if (chol) {  // K C Sigma = Y^T Y
            syrkLowerMinus(n, m, Yt.d.data(), m, Sigma.d.data(), n);
        } else {
            const Mat KCS = gemm(gemm(K, Cb), Sigma);
            for (size_t k = 0; k < Sigma.d.size(); ++k) Sigma.d[k] -= KCS.d[k];
        }
    }
};

void packGroup(const Group& X, double* out) {  // A.q(4) A.x(3) w(3) then per landmark q(4) a(1)
    out[0] = X.A.q.w; out[1] = X.A.q.x; out[2] = X.A.q.y; out[3] = X.A.q.z;
    out[4] = X.A.x.x; out[5] = X.A.x.y; out[6] = X.A.x.z;
    out[7] = X.w.x; out[8] = X.w.y; out[9] = X.w.z;
    for (size_t i = 0; i < X.Q.size(); ++i) {
        double* o = out + 10 + 5 * i;
        o[0] = X.Q[i].q.w; o[1] = X.Q[i].q.x; o[2] = X.Q[i].q.y; o[3] = X.Q[i].q.z; o[4] = X.Q[i].a;
    }
}
void packState(const State& S, double* out) {  // pose.q(4) pose.x(3) vel(3) then per landmark p(3)
    out[0] = S.pose.q.w; out[1] = S.pose.q.x; out[2] = S.pose.q.y; out[3] = S.pose.q.z;
    out[4] = S.pose.x.x; out[5] = S.pose.x.y; out[6] = S.pose.x.z;
    out[7] = S.velocity.x; out[8] = S.velocity.y; out[9] = S.velocity.z;
    for (size_t i = 0; i < S.p.size(); ++i) {
        out[10 + 3 * i] = S.p[i].x; out[11 + 3 * i] = S.p[i].y; out[12 + 3 * i] = S.p[i].z;
    }
}
Group unpackGroup(const double* in, int N) {
    Group X;
    X.A.q = {in[0], in[1], in[2], in[3]};
    X.A.x = {in[4], in[5], in[6]};
    X.w = {in[7], in[8], in[9]};
    X.Q.resize(N);
    X.id.resize(N);
    for (int i = 0; i < N; ++i) {
        const double* o = in + 10 + 5 * i;
        X.Q[i].q = {o[0], o[1], o[2], o[3]};
        X.Q[i].a = o[4];
        X.id[i] = i;
    }
    return X;
}
State unpackState(const double* in, int N, const double* camq, const double* camx) {
    State S;
    S.pose.q = {in[0], in[1], in[2], in[3]};
    S.pose.x = {in[4], in[5], in[6]};
    S.velocity = {in[7], in[8], in[9]};
    S.p.resize(N);
    S.id.resize(N);
    for (int i = 0; i < N; ++i) {
        S.p[i] = {in[10 + 3 * i], in[11 + 3 * i], in[12 + 3 * i]};
        S.id[i] = i;
    }
    S.cameraOffset.q = {camq[0], camq[1], camq[2], camq[3]};
    S.cameraOffset.x = {camx[0], camx[1], camx[2]};
    return S;
}
}  // namespace

// ------------------------------------------------------------------ C interface (ctypes)
extern "C" {
struct oracle_settings {  // must match Settings above field for field
    double v[15];
    int flags[4];
    double initialAccelBias[3], initialOmegaBias[3], cameraOffset_x[3], cameraOffset_q[4];
};
static Settings toSettings(const oracle_settings* o) {
    Settings s;
    double* f[15] = {&s.biasOmegaProcessVariance, &s.biasAccelProcessVariance, &s.gravityProcessVariance,
        &s.velocityProcessVariance, &s.pointProcessVariance, &s.velOmegaVariance, &s.velAccelVariance,
        &s.measurementVariance, &s.initialGravityVariance, &s.initialVelocityVariance, &s.initialPointVariance,
        &s.initialBiasOmegaVariance, &s.initialBiasAccelVariance, &s.initialSceneDepth, &s.outlierThreshold};
    for (int i = 0; i < 15; ++i) *f[i] = o->v[i];
    s.useInnovationLift = o->flags[0];
    s.useDiscreteInnovationLift = o->flags[1];
    s.useDiscreteVelocityLift = o->flags[2];
    s.fastRiccati = o->flags[3];
    std::memcpy(s.initialAccelBias, o->initialAccelBias, sizeof(double) * 3);
    std::memcpy(s.initialOmegaBias, o->initialOmegaBias, sizeof(double) * 3);
    std::memcpy(s.cameraOffset_x, o->cameraOffset_x, sizeof(double) * 3);
    std::memcpy(s.cameraOffset_q, o->cameraOffset_q, sizeof(double) * 4);
    return s;
}

void* oracle_create(const oracle_settings* o) { return new Filter(toSettings(o)); }
void oracle_destroy(void* h) { delete static_cast<Filter*>(h); }
// 0 = dense reference operation sequence (default), 1 = structured backend (same equations, see "structured backend")
void oracle_set_structured(void* h, int on) { static_cast<Filter*>(h)->structured = on != 0; }
// returns 0 ok, -1 antipodal domain_error (libs/core/src/SO3.cpp:160-161)
int oracle_process_imu(void* h, double stamp, const double* w, const double* a) {
    try {
        static_cast<Filter*>(h)->processIMUData(IMU{stamp, {w[0], w[1], w[2]}, {a[0], a[1], a[2]}});
    } catch (const Antipodal&) {
        return -1;
    }
    return 0;
}
int oracle_process_vision(void* h, double stamp, int nb, const int* ids, const double* y) {
    try {
        static_cast<Filter*>(h)->processVisionData(stamp, nb, ids, y);
    } catch (const Antipodal&) {
        return -1;
    }
    return 0;
}
int oracle_num_landmarks(void* h) { return int(static_cast<Filter*>(h)->xi0.p.size()); }
double oracle_get_time(void* h) { return static_cast<Filter*>(h)->currentTime; }
void oracle_get_ids(void* h, int* ids) {
    auto* f = static_cast<Filter*>(h);
    std::copy(f->X.id.begin(), f->X.id.end(), ids);
}
void oracle_get_sigma(void* h, double* out) {
    auto* f = static_cast<Filter*>(h);
    std::copy(f->Sigma.d.begin(), f->Sigma.d.end(), out);
}
void oracle_set_sigma(void* h, const double* in) {
    auto* f = static_cast<Filter*>(h);
    std::copy(in, in + f->Sigma.d.size(), f->Sigma.d.begin());
}
// State injection for single-step (kernel-level) parity tests: overwrites the whole filter with N landmarks `ids`, origin
// state / group element packed as packState / packGroup, bias, Sigma ((11+3N)^2 row-major) and the integrator scalars.
void oracle_set_state(void* h, int N, const int* ids, const double* state, const double* group, const double* bias6,
    const double* sigma, double currentTime, const double* curVel6, const double* accVel6, double accumulatedTime, int initialised) {
    auto* f = static_cast<Filter*>(h);
    const SE3 cam = f->xi0.cameraOffset;
    const double camq[4] = {cam.q.w, cam.q.x, cam.q.y, cam.q.z}, camx[3] = {cam.x.x, cam.x.y, cam.x.z};
    f->xi0 = unpackState(state, N, camq, camx);
    f->X = unpackGroup(group, N);
    for (int i = 0; i < N; ++i) f->xi0.id[i] = f->X.id[i] = ids[i];
    std::memcpy(f->inputBias, bias6, 6 * sizeof(double));
    const int n = SIGMA_BASE_SIZE + 3 * N;
    f->Sigma = Mat(n, n);
    std::copy(sigma, sigma + size_t(n) * n, f->Sigma.d.begin());
    f->currentTime = currentTime;
    f->currentVelocity = IMU{0, {curVel6[0], curVel6[1], curVel6[2]}, {curVel6[3], curVel6[4], curVel6[5]}};
    f->accumulatedVelocity = IMU{0, {accVel6[0], accVel6[1], accVel6[2]}, {accVel6[3], accVel6[4], accVel6[5]}};
    f->accumulatedTime = accumulatedTime;
    f->initialised = initialised != 0;
}
void oracle_get_bias(void* h, double* out) { std::memcpy(out, static_cast<Filter*>(h)->inputBias, 6 * sizeof(double)); }
void oracle_get_xi0(void* h, double* out) { packState(static_cast<Filter*>(h)->xi0, out); }
void oracle_get_estimate(void* h, double* out) { packState(static_cast<Filter*>(h)->stateEstimate(), out); }
void oracle_get_group(void* h, double* out) { packGroup(static_cast<Filter*>(h)->X, out); }
// last-update internals; return their lengths (0 when no update ran)
int oracle_get_last(void* h, double* delta, double* gamma, double* gammaTotal) {
    auto* f = static_cast<Filter*>(h);
    std::copy(f->lastDelta.begin(), f->lastDelta.end(), delta);
    std::copy(f->lastGamma.begin(), f->lastGamma.end(), gamma);
    std::copy(f->lastGammaTotal.begin(), f->lastGammaTotal.end(), gammaTotal);
    return int(f->lastDelta.size());
}

// Stand-alone EqF matrices for block-level parity: packed group/state as packGroup/packState.
// A0 out (5+3N)^2, B out (5+3N)x6, C0 out 2N x (5+3N), all row-major.
int oracle_matrices(int N, const double* group, const double* state, const double* camq, const double* camx,
    const double* omega, double* A0, double* B, double* C0) {
    try {
        const Group X = unpackGroup(group, N);
        const State xi0 = unpackState(state, N, camq, camx);
        const Manifold m = projectToManifold(xi0);
        const Mat A = EqFStateMatrixA(X, m, IMU{0, {omega[0], omega[1], omega[2]}, {}});
        const Mat Bm = EqFInputMatrixB(X, m);
        const Mat C = EqFOutputMatrixC(m);
        std::copy(A.d.begin(), A.d.end(), A0);
        std::copy(Bm.d.begin(), Bm.d.end(), B);
        std::copy(C.d.begin(), C.d.end(), C0);
    } catch (const Antipodal&) {
        return -1;
    }
    return 0;
}
int oracle_bundle_lift(int N, const double* group, const double* state, const double* camq, const double* camx,
    const double* base, const double* SigmaE, double* out) {
    const Group X = unpackGroup(group, N);
    const State xi0 = unpackState(state, N, camq, camx);
    Mat Se(5 + 3 * N, 5 + 3 * N);
    std::copy(SigmaE, SigmaE + Se.d.size(), Se.d.begin());
    const std::vector<double> G = bundleLift(std::vector<double>(base, base + 5 + 3 * N), xi0, X, Se);
    std::copy(G.begin(), G.end(), out);
    return 0;
}
}
