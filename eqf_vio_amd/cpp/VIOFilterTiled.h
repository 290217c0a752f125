// C++ host facade of the 2-D block-partitioned filter (BASELINE configs[4]: one filter of N = 4000 landmarks whose Sigma is spread over
// the GPUs of a node): the reference's VIOFilter interface (eqf_vio/include/eqf_vio/VIOFilter.h:41-88) over the eqf_tf_* entry points
// of include/eqf_vio_amd.h.  One VIOFilterTiled per rank of a Pr x Pc process grid (one process per GPU); every rank makes the same
// calls with the same arguments.  The filter's host loop -- landmark bookkeeping on slots, IMU bursts, the two distributed Cholesky
// factorisations of an update, the downdate -- is behind the C ABI (csrc/eqf_tiledf.hip); this header is plumbing: value types, status
// codes back into the reference's void-or-throw behaviour, and the ONE callback the loop needs from its host, a broadcast of device
// memory along a process row / column.  With RCCL (INTEGRATION.md has the full listing):
//
//     struct Comms { ncclComm_t row[2], col[2], all; };          // one row / column communicator per chain of an update
//     int bcast(void* ctx, int group, int chain, int root, void* buf, size_t bytes, void* stream) {
//         auto* c = static_cast<Comms*>(ctx);
//         ncclComm_t comm = group == 0 ? c->row[chain] : (group == 1 ? c->col[chain] : c->all);
//         return ncclBroadcast(buf, buf, bytes, ncclChar, root, comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
//     }
//
// (root is the sender's index INSIDE the group: its process column for a row broadcast, its process row for a column broadcast -- the
// rank order ncclCommSplit gives when the key is the global rank.)  A 1 x 1 grid needs no callback.  No CPU fallback.
#pragma once
#include "VIOFilter.h"

namespace eqf_vio_amd {

class VIOFilterTiled {
  public:
    using Settings = VIOFilter::Settings;
    std::unique_ptr<Settings> settings;

    // capacity = most landmarks ever tracked at once; blockLandmarks = landmarks per block of the 2-D partition (250 at N = 4000);
    // grid Pr x Pc with Pr | Pc (1 x 1, 1 x 2, 2 x 2, 2 x 4 for one 8-GPU node), rank = pr * Pc + pc; comm may be null on a 1 x 1 grid.
    VIOFilterTiled(const Settings& s, int capacity, int blockLandmarks, int Pr = 1, int Pc = 1, int rank = 0, int device = 0,
        const eqf_tf_comm* comm = nullptr, int reserveCUs = -1)
        : settings(std::make_unique<Settings>(s)) {
        eqf_tf* h = nullptr;
        const int rc = eqf_tf_create(settings.get(), capacity, blockLandmarks, Pr, Pc, rank, device, reserveCUs, comm, &h);
        if (rc != EQF_OK) throw std::runtime_error("eqf_tf_create failed with status " + std::to_string(rc) + " (no CPU fallback)");
        handle_.reset(h);
    }
    VIOFilterTiled(VIOFilterTiled&&) = default;
    VIOFilterTiled& operator=(VIOFilterTiled&&) = default;

    // VIOFilter::processIMUData (VIOFilter.cpp:120-131): void; the reference's silent early-outs stay silent
    void processIMUData(const IMUVelocity& imuVelocity) {
        check(eqf_tf_process_imu(handle_.get(), imuVelocity.stamp, imuVelocity.omega.data(), imuVelocity.accel.data()), "eqf_tf_process_imu");
    }
    // VIOFilter::processVisionData (VIOFilter.cpp:232-302): bearings sorted by ascending id (:239-240)
    void processVisionData(const VisionMeasurement& measurement) {
        const int n = int(measurement.bearings.size());
        std::vector<int> ids(n);
        std::vector<double> y(size_t(3) * n);
        for (int i = 0; i < n; ++i) {
            ids[i] = measurement.bearings[i].id;
            for (int c = 0; c < 3; ++c) y[3 * i + c] = measurement.bearings[i].p[c];
        }
        check(eqf_tf_process_vision(handle_.get(), measurement.stamp, n, ids.data(), y.data()), "eqf_tf_process_vision");
    }
    double getTime() const {  // VIOFilter.cpp:343
        double t = 0;
        check(eqf_tf_get_time(handle_.get(), &t), "eqf_tf_get_time");
        return t;
    }
    VIOState stateEstimate() const {  // VIOFilter.cpp:304: landmarks in the reference's order
        VIOState s;
        const int n = eqf_tf_num_landmarks(handle_.get());
        std::vector<int> ids(std::max(n, 1));
        std::vector<double> p(size_t(3) * std::max(n, 1));
        double q[4], x[3], v[3];
        check(eqf_tf_get_ids(handle_.get(), ids.data(), nullptr), "eqf_tf_get_ids");
        check(eqf_tf_get_state_estimate(handle_.get(), q, x, v, p.data()), "eqf_tf_get_state_estimate");
        s.pose.R = {q[0], q[1], q[2], q[3]};
        s.pose.x = {x[0], x[1], x[2]};
        s.velocity = {v[0], v[1], v[2]};
        s.bodyLandmarks.resize(n);
        for (int i = 0; i < n; ++i) {
            s.bodyLandmarks[i].p = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
            s.bodyLandmarks[i].id = ids[i];
        }
        s.cameraOffset = settings->cameraOffsetSE3();
        return s;
    }
    // VIOFilter::stateCovariance (:306-309): the dense matrix gathered to this rank -- COLLECTIVE, every rank of the grid calls it
    MatrixXd stateCovariance() const {
        MatrixXd S;
        S.n = 11 + 3 * eqf_tf_num_landmarks(handle_.get());
        S.data.assign(size_t(S.n) * S.n, 0.0);
        check(eqf_tf_get_sigma(handle_.get(), S.data.data(), S.n, 0), "eqf_tf_get_sigma");
        return S;
    }
    std::array<double, 6> inputBias() const {
        std::array<double, 6> b{};
        check(eqf_tf_get_bias(handle_.get(), b.data()), "eqf_tf_get_bias");
        return b;
    }
    void synchronize() { check(eqf_tf_synchronize(handle_.get()), "eqf_tf_synchronize"); }
    eqf_tf* handle() { return handle_.get(); }

  private:
    struct Deleter {
        void operator()(eqf_tf* h) const { eqf_tf_destroy(h); }
    };
    std::unique_ptr<eqf_tf, Deleter> handle_;
    void check(int rc, const char* what) const {
        if (rc >= 0) return;  // EQF_OK or one of the reference's silent early-outs
        if (rc == EQF_ERR_NUMERIC) throw std::domain_error(std::string(what) + ": " + eqf_tf_last_error(handle_.get()));
        throw std::runtime_error(std::string(what) + " failed with status " + std::to_string(rc) + " " + eqf_tf_last_error(handle_.get()));
    }
};

}  // namespace eqf_vio_amd
