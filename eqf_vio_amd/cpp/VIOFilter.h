// C++ host facade of the MI355X EqF path: the reference's VIOFilter interface over the C ABI.
//
// Mirrors eqf_vio/include/eqf_vio/VIOFilter.h:41-88 (same class and member names, same argument meaning,
// same silent early-outs) so that a caller of the reference -- eqf_vio/src/main.cpp:86,116,129,134,
// eqf_vio_ros/src/eqf_vio_ros_node.cpp:59,90,104 -- compiles against this header after replacing the Eigen
// value types by the plain arrays below (Eigen is not a dependency of this library).  All arithmetic runs in
// the HIP kernels behind include/eqf_vio_amd.h; this header is plumbing only and has no CPU fallback.
#pragma once
#include <array>
#include <cmath>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/eqf_vio_amd.h"

namespace eqf_vio_amd {

using Vector3d = std::array<double, 3>;
struct Quaterniond {  // Eigen::Quaterniond order of the reference's CSV output: w, x, y, z
    double w = 1, x = 0, y = 0, z = 0;
};
struct SE3 {  // libs/core/include/SE3.h: rotation (quaternion backed, SO3.cpp:25) + translation
    Quaterniond R;
    Vector3d x{0, 0, 0};
};

constexpr double GRAVITY_CONSTANT = 9.81;  // eqf_vio/include/eqf_vio/IMUVelocity.h:22

struct IMUVelocity {  // eqf_vio/include/eqf_vio/IMUVelocity.h:24-37
    double stamp = 0;
    Vector3d omega{0, 0, 0};
    Vector3d accel{0, 0, 0};
};

struct Point3d {  // eqf_vio/include/eqf_vio/VIOState.h:38-41
    Vector3d p{0, 0, 0};
    int id = -1;
};

struct VisionMeasurement {  // eqf_vio/include/eqf_vio/VisionMeasurement.h:24-28 (bearings sorted by ascending id)
    double stamp = 0;
    int numberOfBearings = 0;
    std::vector<Point3d> bearings;
};

struct VIOState {  // eqf_vio/include/eqf_vio/VIOState.h:51-60
    SE3 pose;
    Vector3d velocity{0, 0, 0};
    std::vector<Point3d> bodyLandmarks;
    SE3 cameraOffset;
};

struct SOT3 {  // libs/core/include/SOT3.h
    Quaterniond R;
    double a = 1;
};
struct VIOGroup {  // eqf_vio/include/eqf_vio/VIOGroup.h:24-33
    SE3 A;
    Vector3d w{0, 0, 0};
    std::vector<SOT3> Q;
    std::vector<int> id;
};

// Dense row-major covariance returned by stateCovariance() (the reference returns Eigen::MatrixXd).
struct MatrixXd {
    int n = 0;
    std::vector<double> data;
    double operator()(int i, int j) const { return data[size_t(i) * n + j]; }
    int rows() const { return n; }
    int cols() const { return n; }
};

struct AuxiliaryFilterData {  // eqf_vio/include/eqf_vio/VIOFilter.h:30-39
    Quaterniond initialAttitude;
    Vector3d initialPosition{0, 0, 0};
    double initialTime = 0;
    double measurementVariance = 0.1;
    double processVariance = 1.0;
    double omegaVariance = 0.1;
    double accelVariance = 0.1;
    SE3 cameraOffset;
};

class VIOFilter {
  public:
    // eqf_vio/include/eqf_vio/VIOFilterSettings.h:28-54.  YAML parsing is host plumbing outside this path: fill
    // the fields directly (same names and defaults as the reference).
    struct Settings : eqf_settings {
        Settings() { eqf_settings_default(this); }
        SE3 cameraOffsetSE3() const {
            SE3 T;
            T.R = {cameraOffset_q[0], cameraOffset_q[1], cameraOffset_q[2], cameraOffset_q[3]};
            T.x = {cameraOffset_x[0], cameraOffset_x[1], cameraOffset_x[2]};
            return T;
        }
    };
    std::unique_ptr<Settings> settings;

    // VIOFilter(const VIOFilter::Settings&) (VIOFilter.cpp:60-73).  capacity = most landmarks ever tracked at once
    // (the reference grows Sigma on demand; device buffers are sized once).
    explicit VIOFilter(const Settings& s, int capacity = 256, int device = 0, int precision = EQF_PRECISION_F64)
        : settings(std::make_unique<Settings>(s)) {
        eqf_filter* h = nullptr;
        const int rc = eqf_create(settings.get(), capacity, 1, device, precision, &h);
        if (rc != EQF_OK) throw std::runtime_error("eqf_create failed with status " + std::to_string(rc) + " (no CPU fallback)");
        handle_.reset(h);
    }
    // VIOFilter(const AuxiliaryFilterData&, const VIOFilter::Settings&) (VIOFilter.cpp:51-58)
    VIOFilter(const AuxiliaryFilterData& auxiliaryData, const Settings& s, int capacity = 256, int device = 0,
        int precision = EQF_PRECISION_F64)
        : VIOFilter(s, capacity, device, precision) {
        setAuxiliaryData(auxiliaryData);
    }
    VIOFilter(VIOFilter&&) = default;             // move-only, like the reference (unique_ptr settings)
    VIOFilter& operator=(VIOFilter&&) = default;  // eqf_vio_ros_node.cpp:59 move-assigns

    void reset() { check(eqf_reset(handle_.get()), "eqf_reset"); }  // VIOFilter.cpp:84-91

    // VIOFilter.cpp:133-144: origin pose = identity position + the attitude that takes the measured specific-force
    // direction to e3, zero origin velocity, initialised.  processIMUData does this lazily on the device at the first
    // sample (with the bias-corrected sample); this member is the explicit form of the reference's public method and takes
    // the velocity as given.  Throws std::domain_error like SO3FromVectors (SO3.cpp:160-161) for accel = -|accel| e3.
    void initialiseFromIMUData(const IMUVelocity& imuVelocity) {
        Snapshot st = dump();
        st.pq = so3FromVectors(imuVelocity.accel, {0, 0, 1});
        st.px = {0, 0, 0};
        st.v = {0, 0, 0};
        st.initialised = 1;
        restore(st);
    }

    // VIOFilter.cpp:120-131
    void processIMUData(const IMUVelocity& imuVelocity) {
        check(eqf_process_imu(handle_.get(), &imuVelocity.stamp, imuVelocity.omega.data(), imuVelocity.accel.data(), &lastStatus_),
            "eqf_process_imu");
    }
    // VIOFilter.cpp:232-302.  Throws std::domain_error where the reference's SO3FromVectors would (SO3.cpp:160).
    void processVisionData(const VisionMeasurement& measurement) {
        const int nb = int(measurement.bearings.size());
        ids_.resize(nb);
        y_.resize(size_t(3) * nb);
        for (int i = 0; i < nb; ++i) {
            ids_[i] = measurement.bearings[i].id;
            for (int c = 0; c < 3; ++c) y_[size_t(3) * i + c] = measurement.bearings[i].p[c];
        }
        check(eqf_process_vision(handle_.get(), &measurement.stamp, &nb, ids_.data(), y_.data(), nb, &lastStatus_), "eqf_process_vision");
    }

    // VIOFilter.cpp:74-82: origin pose from the given attitude / position, zero origin velocity, camera offset replaced,
    // filter marked initialised (so the first IMU sample does no gravity alignment).
    void setAuxiliaryData(const AuxiliaryFilterData& auxiliaryData) {
        auxData = auxiliaryData;
        Snapshot st = dump();
        st.pq = {auxiliaryData.initialAttitude.w, auxiliaryData.initialAttitude.x, auxiliaryData.initialAttitude.y, auxiliaryData.initialAttitude.z};
        st.px = auxiliaryData.initialPosition;
        st.v = {0, 0, 0};
        st.initialised = 1;
        const SE3& T = auxiliaryData.cameraOffset;
        const double cq[4] = {T.R.w, T.R.x, T.R.y, T.R.z};
        check(eqf_set_camera_offset(handle_.get(), cq, T.x.data()), "eqf_set_camera_offset");
        for (int i = 0; i < 4; ++i) settings->cameraOffset_q[i] = cq[i];
        for (int i = 0; i < 3; ++i) settings->cameraOffset_x[i] = T.x[i];
        restore(st);
    }
    // VIOFilter.cpp:93-118: the landmark set becomes the given inertial-frame points (ids as given), Q_i = identity,
    // Sigma = initialPointVariance * I outside the 11 x 11 base block.
    void setInertialPoints(const std::vector<Point3d>& inertialPoints) {
        const int N = int(inertialPoints.size());
        Snapshot st = dump();
        // inertialToCameraTF = (xi0.pose * xi0.cameraOffset)^-1
        const double* cq = settings->cameraOffset_q;
        const double* cx = settings->cameraOffset_x;
        const std::array<double, 4> tq = qmul(st.pq, {cq[0], cq[1], cq[2], cq[3]});
        const Vector3d rx = qrot(st.pq, {cx[0], cx[1], cx[2]});
        const Vector3d tx = {st.px[0] + rx[0], st.px[1] + rx[1], st.px[2] + rx[2]};
        const std::array<double, 4> tqi = {tq[0], -tq[1], -tq[2], -tq[3]};
        st.ids.resize(N);
        st.p0.assign(size_t(3) * N, 0.0);
        st.Qq.assign(size_t(4) * N, 0.0);
        st.Qa.assign(N, 1.0);
        for (int i = 0; i < N; ++i) {
            const Vector3d& p = inertialPoints[i].p;
            const Vector3d q = qrot(tqi, {p[0] - tx[0], p[1] - tx[1], p[2] - tx[2]});
            st.ids[i] = inertialPoints[i].id;
            for (int c = 0; c < 3; ++c) st.p0[size_t(3) * i + c] = q[c];
            st.Qq[size_t(4) * i] = 1.0;
        }
        const int n = 11 + 3 * N;
        std::vector<double> S(size_t(n) * n, 0.0);
        for (int i = 0; i < n; ++i) S[size_t(i) * n + i] = settings->initialPointVariance;
        for (int r = 0; r < 11; ++r)
            for (int c = 0; c < 11; ++c) S[size_t(r) * n + c] = st.sigma[size_t(r) * st.n + c];
        st.sigma.swap(S);
        st.n = n;
        restore(st);
    }

    double getTime() const {  // VIOFilter.cpp:343
        double t = 0;
        eqf_get_time(handle_.get(), &t);
        return t;
    }
    VIOState stateEstimate() const {  // VIOFilter.cpp:304
        VIOState s;
        const int N = eqf_num_landmarks(handle_.get(), 0);
        std::vector<double> p(size_t(3) * (N > 0 ? N : 1));
        std::vector<int> ids(N > 0 ? N : 1);
        double q[4], x[3], v[3];
        check(eqf_get_state_estimate(handle_.get(), 0, q, x, v, p.data()), "eqf_get_state_estimate");
        check(eqf_get_ids(handle_.get(), 0, ids.data()), "eqf_get_ids");
        s.pose.R = {q[0], q[1], q[2], q[3]};
        s.pose.x = {x[0], x[1], x[2]};
        s.velocity = {v[0], v[1], v[2]};
        s.bodyLandmarks.resize(N);
        for (int i = 0; i < N; ++i) {
            s.bodyLandmarks[i].p = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
            s.bodyLandmarks[i].id = ids[i];
        }
        s.cameraOffset = settings->cameraOffsetSE3();
        check(eqf_device_error(handle_.get()) == 0 ? 0 : EQF_ERR_NUMERIC, "device");
        return s;
    }
    MatrixXd stateCovariance() const {  // VIOFilter.cpp:306-309
        MatrixXd S;
        S.n = 11 + 3 * eqf_num_landmarks(handle_.get(), 0);
        S.data.resize(size_t(S.n) * S.n);
        check(eqf_get_sigma(handle_.get(), 0, S.data.data(), S.n), "eqf_get_sigma");
        return S;
    }
    int lastStatus() const { return lastStatus_; }  // EQF_SKIPPED_* where the reference returns early
    eqf_filter* handle() const { return handle_.get(); }

    // operator<<(ostream&, const VIOFilter&) (VIOFilter.cpp:311-341): xi0, X, N, per-landmark (id, p0, Q, a), Sigma
    friend std::ostream& operator<<(std::ostream& os, const VIOFilter& f) {
        const int N = eqf_num_landmarks(f.handle_.get(), 0);
        std::vector<double> p(size_t(3) * (N > 0 ? N : 1)), Qq(size_t(4) * (N > 0 ? N : 1)), Qa(N > 0 ? N : 1);
        std::vector<int> ids(N > 0 ? N : 1);
        double q[4], x[3], v[3], Aq[4], Ax[3], w[3];
        eqf_get_origin(f.handle_.get(), 0, q, x, v, p.data());
        eqf_get_group(f.handle_.get(), 0, Aq, Ax, w, Qq.data(), Qa.data());
        eqf_get_ids(f.handle_.get(), 0, ids.data());
        os << x[0] << ", " << x[1] << ", " << x[2] << ", " << q[0] << ", " << q[1] << ", " << q[2] << ", " << q[3] << ", ";
        os << v[0] << ", " << v[1] << ", " << v[2] << ", ";
        os << Ax[0] << ", " << Ax[1] << ", " << Ax[2] << ", " << Aq[0] << ", " << Aq[1] << ", " << Aq[2] << ", " << Aq[3] << ", ";
        os << w[0] << ", " << w[1] << ", " << w[2] << ", " << N;
        for (int i = 0; i < N; ++i) {
            os << ", " << ids[i] << ", " << p[3 * i] << ", " << p[3 * i + 1] << ", " << p[3 * i + 2];
            os << ", " << Qq[4 * i] << ", " << Qq[4 * i + 1] << ", " << Qq[4 * i + 2] << ", " << Qq[4 * i + 3] << ", " << Qa[i];
        }
        const MatrixXd S = f.stateCovariance();
        for (double s : S.data) os << ", " << s;
        return os;
    }

    AuxiliaryFilterData auxData;  // VIOFilter.h:43

  private:
    // full-precision snapshot of the filter (what eqf_set_state takes)
    struct Snapshot {
        std::vector<int> ids;
        std::array<double, 4> pq{1, 0, 0, 0}, Aq{1, 0, 0, 0};
        Vector3d px{0, 0, 0}, v{0, 0, 0}, Ax{0, 0, 0}, w{0, 0, 0};
        std::vector<double> p0, Qq, Qa, sigma;
        double bias[6], curVel[6], accVel[6], accTime = 0, time = -1;
        int initialised = 0, n = 11;
    };
    Snapshot dump() const {
        Snapshot st;
        eqf_filter* h = handle_.get();
        const int N = eqf_num_landmarks(h, 0);
        const int M = N > 0 ? N : 1;
        st.ids.resize(M);
        st.p0.resize(size_t(3) * M);
        st.Qq.resize(size_t(4) * M);
        st.Qa.resize(M);
        check(eqf_get_ids(h, 0, st.ids.data()), "eqf_get_ids");
        check(eqf_get_origin(h, 0, st.pq.data(), st.px.data(), st.v.data(), st.p0.data()), "eqf_get_origin");
        check(eqf_get_group(h, 0, st.Aq.data(), st.Ax.data(), st.w.data(), st.Qq.data(), st.Qa.data()), "eqf_get_group");
        check(eqf_get_bias(h, 0, st.bias), "eqf_get_bias");
        check(eqf_get_integrator(h, 0, st.curVel, st.accVel, &st.accTime, &st.initialised), "eqf_get_integrator");
        check(eqf_get_time(h, &st.time), "eqf_get_time");
        st.ids.resize(N);
        st.n = 11 + 3 * N;
        st.sigma.resize(size_t(st.n) * st.n);
        check(eqf_get_sigma(h, 0, st.sigma.data(), st.n), "eqf_get_sigma");
        return st;
    }
    void restore(const Snapshot& st) {
        const int N = int(st.ids.size());
        check(eqf_set_state(handle_.get(), 0, N, st.ids.data(), st.pq.data(), st.px.data(), st.v.data(), st.p0.data(), st.Aq.data(),
                  st.Ax.data(), st.w.data(), st.Qq.data(), st.Qa.data(), st.bias, st.sigma.data(), st.n, st.time, st.curVel, st.accVel,
                  st.accTime, st.initialised),
            "eqf_set_state");
    }
    static std::array<double, 4> qmul(const std::array<double, 4>& a, const std::array<double, 4>& b) {
        return {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3], a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]};
    }
    // SO3::SO3FromVectors (libs/core/src/SO3.cpp:155-167): R = I + v^ + v^ v^ / (1 + c) for the normalised inputs, stored as
    // a quaternion the way Eigen converts a rotation matrix (SO3.cpp:100).
    static std::array<double, 4> so3FromVectors(const Vector3d& origin, const Vector3d& dest) {
        auto unit = [](const Vector3d& a) {
            const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
            return Vector3d{a[0] / n, a[1] / n, a[2] / n};
        };
        const Vector3d o = unit(origin), d = unit(dest);
        const Vector3d v = {o[1] * d[2] - o[2] * d[1], o[2] * d[0] - o[0] * d[2], o[0] * d[1] - o[1] * d[0]};
        const double c = o[0] * d[0] + o[1] * d[1] + o[2] * d[2];
        if (std::fabs(1 + c) <= 1e-8) throw std::domain_error("The vectors cannot be exactly opposing.");
        const double k = 1 / (1 + c);
        double m[3][3];
        const double vx[3][3] = {{0, -v[2], v[1]}, {v[2], 0, -v[0]}, {-v[1], v[0], 0}};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double sq = 0;
                for (int l = 0; l < 3; ++l) sq += vx[i][l] * vx[l][j];
                m[i][j] = (i == j ? 1.0 : 0.0) + vx[i][j] + k * sq;
            }
        std::array<double, 4> q{};
        double t = m[0][0] + m[1][1] + m[2][2];
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
            const int j = (i + 1) % 3, l = (j + 1) % 3;
            t = std::sqrt(m[i][i] - m[j][j] - m[l][l] + 1.0);
            q[1 + i] = 0.5 * t;
            t = 0.5 / t;
            q[0] = (m[l][j] - m[j][l]) * t;
            q[1 + j] = (m[j][i] + m[i][j]) * t;
            q[1 + l] = (m[l][i] + m[i][l]) * t;
        }
        return q;
    }
    static Vector3d qrot(const std::array<double, 4>& q, const Vector3d& v) {
        const Vector3d u = {q[1], q[2], q[3]};
        const Vector3d t = {2 * (u[1] * v[2] - u[2] * v[1]), 2 * (u[2] * v[0] - u[0] * v[2]), 2 * (u[0] * v[1] - u[1] * v[0])};
        return {v[0] + q[0] * t[0] + u[1] * t[2] - u[2] * t[1], v[1] + q[0] * t[1] + u[2] * t[0] - u[0] * t[2],
            v[2] + q[0] * t[2] + u[0] * t[1] - u[1] * t[0]};
    }
    struct Deleter {
        void operator()(eqf_filter* h) const { eqf_destroy(h); }
    };
    static void check(int rc, const char* what) {
        if (rc == EQF_ERR_NUMERIC) throw std::domain_error("The vectors cannot be exactly opposing.");  // SO3.cpp:160-161
        if (rc < 0) throw std::runtime_error(std::string(what) + " failed with status " + std::to_string(rc));
    }
    std::unique_ptr<eqf_filter, Deleter> handle_;
    std::vector<int> ids_;
    std::vector<double> y_;
    int lastStatus_ = 0;
};

}  // namespace eqf_vio_amd
