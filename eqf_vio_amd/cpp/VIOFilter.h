// C++ host facade of the MI355X EqF path: the reference's VIOFilter interface over the C ABI.
//
// Mirrors eqf_vio/include/eqf_vio/VIOFilter.h:41-88 (same class and member names, same argument meaning,
// same silent early-outs) so that a caller of the reference -- eqf_vio/src/main.cpp:86,116,129,134,
// eqf_vio_ros/src/eqf_vio_ros_node.cpp:59,90,104 -- compiles against this header after replacing the Eigen
// value types by the plain arrays below (Eigen is not a dependency of this library).  All arithmetic runs in
// the HIP kernels behind include/eqf_vio_amd.h; this header is plumbing only and has no CPU fallback.
#pragma once
#include <array>
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
    VIOFilter(VIOFilter&&) = default;             // move-only, like the reference (unique_ptr settings)
    VIOFilter& operator=(VIOFilter&&) = default;  // eqf_vio_ros_node.cpp:59 move-assigns

    void reset() { check(eqf_reset(handle_.get()), "eqf_reset"); }  // VIOFilter.cpp:84-91

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

  private:
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
