/* eqf_vio_amd -- C ABI of the MI355X-native EqF propagate/update hot path.
 *
 * Drop-in boundary for pvangoor/eqf_vio's VIOFilter (eqf_vio/include/eqf_vio/VIOFilter.h:41-88).  The
 * reference has no FFI: the C++ class *is* the boundary, so every entry point below names the member
 * it replaces.  A handle owns a BATCH of `batch` independent filters (batch = 1 for the drop-in
 * case; >1 for Monte-Carlo / multi-sequence runs, BASELINE cfg 4); per-filter arguments are arrays of
 * length `batch`.  One handle = one HIP stream; calls on a handle must be serialised by the caller
 * (the reference is single-threaded too).  Calls enqueue GPU work and return; getters synchronise.
 *
 * Sigma index map (unchanged from the reference, VIOFilter.cpp:54-57,163-167,425):
 *   [0,3) gyro bias  [3,6) accel bias  [6,8) gravity dir  [8,11) velocity  [11+3i,14+3i) landmark i
 *
 * Status codes: 0 ok; >0 "silently skipped" exactly where the reference returns early;
 *               <0 error.  The library NEVER computes on the CPU: without a GPU eqf_create fails.
 */
#ifndef EQF_VIO_AMD_H
#define EQF_VIO_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EQF_OK 0
#define EQF_SKIPPED_BEFORE_FIRST_IMU 1 /* VIOFilter.cpp:147-148 (currentTime < 0)            */
#define EQF_SKIPPED_NONPOSITIVE_DT 2   /* VIOFilter.cpp:150-152, :234-236 (dt <= 0)          */
#define EQF_SKIPPED_NOT_INITIALISED 3  /* VIOFilter.cpp:235 (!initialisedFlag)               */
#define EQF_SKIPPED_NO_BEARINGS 4      /* VIOFilter.cpp:258-259                              */
#define EQF_ERR_INVALID -1
#define EQF_ERR_NO_DEVICE -2
#define EQF_ERR_HIP -3
#define EQF_ERR_CAPACITY -4      /* more landmarks than the handle was created for          */
#define EQF_ERR_UNSORTED -5      /* bearings not sorted by ascending id (VIOFilter.cpp:239)  */
#define EQF_ERR_NUMERIC -6       /* NaN / antipodal SO3FromVectors (SO3.cpp:160) on device   */
#define EQF_ERR_UNSUPPORTED -7

#define EQF_PRECISION_F64 0 /* Sigma stored and contracted in fp64 (parity grade, default)    */
#define EQF_PRECISION_F32 1 /* Sigma stored fp32, propagate + downdate in fp32 (MFMA f32)      */

/* VIOFilter::Settings, eqf_vio/include/eqf_vio/VIOFilterSettings.h:28-54 (same names, same defaults). */
typedef struct eqf_settings {
    double biasOmegaProcessVariance;
    double biasAccelProcessVariance;
    double gravityProcessVariance;
    double velocityProcessVariance;
    double pointProcessVariance;
    double velOmegaVariance;
    double velAccelVariance;
    double measurementVariance;
    double initialGravityVariance;
    double initialVelocityVariance;
    double initialPointVariance;
    double initialBiasOmegaVariance;
    double initialBiasAccelVariance;
    double initialSceneDepth;
    double outlierThreshold;
    int useInnovationLift;
    int useDiscreteInnovationLift;
    int useDiscreteVelocityLift;
    int fastRiccati;
    double initialAccelBias[3];
    double initialOmegaBias[3];
    double cameraOffset_x[3]; /* SE3 cameraOffset: translation ...                             */
    double cameraOffset_q[4]; /* ... and attitude quaternion (w, x, y, z), YAML order "xw"      */
} eqf_settings;

/* Fills the defaults of VIOFilterSettings.h:29-50. */
void eqf_settings_default(eqf_settings* s);

typedef struct eqf_filter eqf_filter; /* opaque */

/* VIOFilter(const Settings&) (VIOFilter.cpp:60-73) for `batch` filters with room for
 * `capacity_landmarks` landmarks each, on HIP device `device`, Sigma precision EQF_PRECISION_*. */
int eqf_create(const eqf_settings* settings, int capacity_landmarks, int batch, int device, int precision,
    eqf_filter** out);
void eqf_destroy(eqf_filter* f);
/* Back to the freshly constructed state (every filter of the batch): no landmarks, identity group element, Sigma and
 * bias from the settings, not initialised, time -1.  Takes the place of VIOFilter::reset() (VIOFilter.cpp:84-91), which
 * no caller in the reference uses and which is deliberately NOT reproduced literally: it also forgets xi0.cameraOffset,
 * sets Sigma to the 11x11 identity and keeps initialisedFlag / inputBias -- with a level identity pose its next Riccati
 * step throws from SO3FromVectors (SO3.cpp:160). */
int eqf_reset(eqf_filter* f);

/* VIOFilter::processIMUData (VIOFilter.cpp:120-131).  stamps[batch], omega[batch][3], accel[batch][3].
 * status (may be NULL) receives one code per filter. */
int eqf_process_imu(eqf_filter* f, const double* stamps, const double* omega, const double* accel, int* status);

/* VIOFilter::processVisionData (VIOFilter.cpp:232-302).  For filter b: nb[b] bearings with ids
 * ids[b*stride + k] (STRICTLY ascending) and unit vectors bearings[(b*stride + k)*3 + c].
 * EQF_ERR_INVALID / EQF_ERR_CAPACITY (nb[b] > capacity) / EQF_ERR_UNSORTED are returned before any effect: the
 * filters are exactly as before the call.  status[] is written whenever the call got past those checks.
 * The call enqueues and returns: the landmark bookkeeping (VIOFilter.cpp:345-443) including the outlier gate is decided and carried out
 * on the device, the update follows in the stream; the handle's id lists (eqf_num_landmarks, eqf_get_ids) are brought up to date from the
 * device's answer when the caller next touches the handle.  One consequence for status[]: a frame in which EVERY landmark of a filter is
 * thrown out as an outlier and none is new reports EQF_OK, not EQF_SKIPPED_NO_BEARINGS (the update is skipped on the device). */
int eqf_process_vision(eqf_filter* f, const double* stamps, const int* nb, const int* ids, const double* bearings,
    int stride, int* status);

/* Stream mode: the whole input stream is made resident in HBM once, then events are replayed by index
 * with no host->device traffic in the loop (how bench.py times the path).
 *   imu:      [K][batch][7]  (stamp, wx, wy, wz, ax, ay, az)      -- IMUVelocity.h:24-37
 *   vstamps:  [F][batch]; ids: [nbear] shared by all frames and filters, ascending;
 *   bearings: [F][batch][nbear][3]                                 -- VisionMeasurement.h:24-28 */
int eqf_stream_upload(eqf_filter* f, int K, const double* imu, int F, const double* vstamps, int nbear,
    const int* ids, const double* bearings);
int eqf_stream_imu(eqf_filter* f, int k);    /* == eqf_process_imu on record k    */
int eqf_stream_vision(eqf_filter* f, int fr); /* == eqf_process_vision on frame fr */

/* Getters (copy to caller memory, synchronise the handle's stream). b = filter index in the batch. */
int eqf_synchronize(eqf_filter* f);
int eqf_get_time(eqf_filter* f, double* t /* [batch] */);               /* VIOFilter::getTime          */
int eqf_num_landmarks(eqf_filter* f, int b);                             /* xi0.bodyLandmarks.size()     */
int eqf_get_ids(eqf_filter* f, int b, int* ids);
/* VIOFilter::stateEstimate (VIOFilter.cpp:304): pose q[4] (w,x,y,z), pose x[3], velocity[3],
 * landmarks p[N][3] (camera frame). */
int eqf_get_state_estimate(eqf_filter* f, int b, double* pose_q, double* pose_x, double* velocity, double* p);
/* Filter internals (what operator<<(VIOFilter) dumps, VIOFilter.cpp:311-341): origin state xi0 ...   */
int eqf_get_origin(eqf_filter* f, int b, double* pose_q, double* pose_x, double* velocity, double* p);
/* ... group element X = (A, w, Q_i = (q_i, a_i)) ...                                                   */
int eqf_get_group(eqf_filter* f, int b, double* A_q, double* A_x, double* w, double* Q_q /*[N][4]*/,
    double* Q_a /*[N]*/);
/* ... and input bias (b_omega, b_accel).                                                               */
int eqf_get_bias(eqf_filter* f, int b, double* bias6);
/* VIOFilter::stateCovariance (VIOFilter.cpp:306-309): n x n, n = 11 + 3N, row-major with leading
 * dimension ld >= n, in the reference's index map. */
int eqf_get_sigma(eqf_filter* f, int b, double* dst, int ld);
/* Test hook: overwrite Sigma (same layout as eqf_get_sigma). */
int eqf_set_sigma(eqf_filter* f, int b, const double* src, int ld);
/* Checkpoint / resume (full precision; the reference's only dump is the lossy, write-only operator<<,
 * VIOFilter.cpp:311-341).  eqf_set_state overwrites filter b with: N landmarks with ids[N]; origin state xi0
 * (pose_q, pose_x, velocity, p0[N][3]); group element X (A_q, A_x, w, Q_q[N][4], Q_a[N]); bias6; Sigma (n x n, n =
 * 11 + 3N, leading dimension ld, reference index map); currentTime; currentVelocity6 (omega, accel) and the
 * accumulated velocity/time of the fastRiccati path; initialised flag.  Everything eqf_get_* returns can be fed back. */
int eqf_set_state(eqf_filter* f, int b, int N, const int* ids, const double* pose_q, const double* pose_x,
    const double* velocity, const double* p0, const double* A_q, const double* A_x, const double* w, const double* Q_q,
    const double* Q_a, const double* bias6, const double* sigma, int ld, double currentTime, const double* currentVelocity6,
    const double* accumulatedVelocity6, double accumulatedTime, int initialised);
/* xi0.cameraOffset = T_IC for the whole batch (VIOFilter::setAuxiliaryData, VIOFilter.cpp:74-82, overwrites the value
 * taken from the settings at construction, :68).  q must be a unit quaternion (w,x,y,z). */
int eqf_set_camera_offset(eqf_filter* f, const double* q, const double* x);
/* The remaining scalar state needed for a lossless dump: currentVelocity6, accumulatedVelocity6, accumulatedTime,
 * initialised flag (any pointer may be NULL). */
int eqf_get_integrator(eqf_filter* f, int b, double* currentVelocity6, double* accumulatedVelocity6, double* accumulatedTime,
    int* initialised);

/* Internals of the most recent update of filter b: delta[2N], gamma[11+3N] (K*delta), Gamma[9+3N]. */
int eqf_get_last_update(eqf_filter* f, int b, double* delta, double* gamma, double* Gamma);
/* Sticky device-side error flag, 0 if none; a bit mask (any bit -> the C++ facade throws std::domain_error, like the reference's
 * SO3FromVectors, SO3.cpp:160): 1 antipodal vectors / singular gravity chart in a propagate step; 2 the same while building the residual
 * or C0i; 4 a pivot of S or Sigma_e not positive; 8 antipodal vectors in the innovation lift (numeric, one filter: the other filters of a
 * batch handle keep running); 16 / 32 a new / restored landmark on the chart pole; 64 singular gravity chart in the dense Riccati backend;
 * 128 an in-launch hand-off of k_chol_resident / k_burst_fused timed out (0.5 s: the GPU was taken away from the launch for that long) --
 * that launch did not write Sigma and the handle must be reset (eqf_reset) or restored (eqf_set_state on EVERY filter of the handle: the call
 * that restores the last one clears bit 128, and bit 4 with it -- a chain that unwinds may have judged pivots of operands it never got): until
 * then every later launch with hand-offs leaves at once.  (After a timeout inside an IMU burst the covariance the handle points at is the other ping-pong
 * buffer, i.e. NOT a valid state: restore, do not continue.)
 * About the hand-offs behind bit 128: k_chol_resident (one launch per vision update, csrc/eqf_resident.hpp) lets workgroups of ONE launch
 * wait for each other.  It is free of deadlock on any grid -- also many times larger than the chip -- under ONE assumption about the
 * hardware that HIP does not document: workgroups are started in the order of their linear index (a workgroup only ever waits for
 * lower indices, so whatever is resident contains a runnable one).  If a device or driver ever broke that order, or took the GPU away
 * for more than 0.5 s (preemption, a debugger), nothing hangs and nothing wrong is written: the first wait that times out sets bit 128 at
 * once, every other wait of the launch sees the bit within microseconds and gives up, no workgroup publishes anything after a failed
 * wait, the covariance downdate does not run (Sigma keeps its pre-update value), and the call that next touches the handle returns
 * EQF_ERR_NUMERIC.  The flag is sticky: later updates of the handle leave at once until eqf_reset.
 * EQF_CHOL_RESIDENT=0 selects one launch per 64-wide block column instead (no in-launch dependency at all).
 * (Fault injection for these hand-offs, the developer toggles, the per-kernel timers and the dense tile kernels behind the partitioned
 * filter are test / measurement hooks: include/eqf_vio_amd_debug.h -- a caller of the reference's interface needs none of them.) */
int eqf_device_error(eqf_filter* f);
/* IMU bursts.  processIMUData calls (VIOFilter.cpp:120-131) only depend on each other and on the state, so the library
 * queues them on the host and launches up to 15 of them -- plus the integrateUpToTime of the processVisionData call
 * that follows (VIOFilter.cpp:233) -- as ONE pair of kernels that reads and writes Sigma once (csrc/eqf_burst.hpp).
 * Each step is the reference's step, in order; the result does not depend on where the bursts are cut.  A queued call
 * is launched when the queue is full, when the next vision call arrives, or when any other entry point of the handle
 * is called (getters, eqf_synchronize, ...): the deferral is invisible apart from timing.  max_steps = 0 launches
 * every call at once through the single-step kernel (k_propagate); default 15 (environment: EQF_IMU_BURST). */
int eqf_set_imu_burst(eqf_filter* f, int max_steps);

/* Propagate backend: 0 = block-structured HBM-bound kernel (default, product path),
 * 1 = dense F Sigma F^T on MFMA (what the reference executes; BASELINE cfg 3 cross-check). */
int eqf_set_dense_propagate(eqf_filter* f, int on);

/* ================================================================================================================================
 * BASELINE configs[4]: ONE filter with N = 4000 landmarks whose Sigma (1.15 GB) is 2-D block-partitioned over the GPUs of a node.
 * One process per GPU; process (pr, pc) of a Pr x Pc grid owns the landmark blocks I = pr, pr + Pr, ... as rows and J = pc, pc + Pc,
 * ... as columns of ONE dense local matrix Sll (3 nlr x 3 nlc doubles, row-major, CALLER-OWNED device memory, e.g. a torch tensor, so
 * that torch.distributed can move pieces of it).  The O(N) filter state and the 11-row base panel Sigma[0:11, :] are REPLICATED: every
 * rank advances its own identical copy inside its eqf_tiled handle, with the same device functions as the single-GPU path.  The
 * exchange schedule (which block row is factored where, the RCCL broadcasts of the solved block rows) is eqf_vio_amd/tiled.py; the
 * entry points below (and the dense tile kernels eqf_tile_* of eqf_vio_amd_debug.h, for a host that writes its own schedule) are what a rank
 * runs between two exchanges.  They enqueue on the handle's stream (eqf_tiled_set_stream; NULL = the
 * default stream) and return; getters synchronise.  fp64 only.
 * Landmark churn (VIOFilter.cpp:345-443) works on SLOTS: the partition is over the handle's N physical landmark slots, a removed
 * landmark leaves an inactive slot behind (zero rows / columns of Sigma with a unit diagonal block, identity linearisation, no
 * measurement rows: a decoupled block that both factorisations carry along as exact zeros), a new landmark takes a free slot -- nothing
 * ever moves between ranks (eqf_tiled_edit_landmarks).  Which landmark sits in which slot, and the reference's landmark ORDER, are the
 * caller's (eqf_vio_amd/tiled.py: TiledFilter); every per-landmark array of the getters below is in SLOT order, holes included.
 * ================================================================================================================================ */
typedef struct eqf_tiled eqf_tiled; /* opaque */

/* VIOFilter(const Settings&) (VIOFilter.cpp:60-73) for the replicated part of one filter with room for capacity_landmarks. */
int eqf_tiled_create(const eqf_settings* settings, int capacity_landmarks, int device, eqf_tiled** out);
void eqf_tiled_destroy(eqf_tiled* t);
int eqf_tiled_set_stream(eqf_tiled* t, void* stream);
/* The rank's share: rowMap[nlr] / colMap[nlc] = global landmark index of every local row / column landmark (host arrays, copied). */
int eqf_tiled_set_geometry(eqf_tiled* t, int nlr, const int* rowMap, int nlc, const int* colMap);

/* VIOFilter::processIMUData (is_imu = 1, VIOFilter.cpp:120-131) or the integrateUpToTime of processVisionData (is_imu = 0, :233;
 * omega / accel ignored): state, base panel and -- when the call does a Riccati step -- the local blocks Sll in place
 * (VIOFilter.cpp:188-189, F = [[F_bb, 0], [L, D]]).  Sll may be NULL while there are no landmarks.  Returns EQF_OK or the
 * EQF_SKIPPED_* code of the reference's silent early-outs. */
int eqf_tiled_propagate(eqf_tiled* t, double stamp, const double* omega, const double* accel, int is_imu, double* Sll, int ldl);
/* K <= 16 consecutive calls of eqf_tiled_propagate as ONE pass over the local blocks: the IMU calls between two vision frames and, with
 * last_is_vision, the vision call's integrateUpToTime as the last one (stamps[K]; omega / accel [K][3], rows of IMU calls only are read).
 * Every call keeps its own linearisation and its own base panel (K small O(N) launches), the Riccati steps of the local blocks are applied
 * back to back in registers (VIOFilter.cpp:188-189 K times; Sll is read and written once).  status[k] = what eqf_tiled_propagate would have
 * returned for call k; the state afterwards is the state after the K calls, call for call. */
int eqf_tiled_propagate_burst(eqf_tiled* t, int K, const double* stamps, const double* omega, const double* accel, int last_is_vision,
    double* Sll, int ldl, int* status);
/* addNewLandmarks on an empty state (VIOFilter.cpp:345-391, :361-366): n landmarks p0 = y * initialSceneDepth, Q = identity;
 * bearings[n][3] host memory.  Sll (geometry set for n landmarks) is initialised: initialPointVariance on the diagonal.
 * EQF_ERR_UNSUPPORTED if the filter already has landmarks. */
int eqf_tiled_add_landmarks(eqf_tiled* t, int n, const double* bearings, double* Sll, int ldl);
/* removeLandmarkAtIndex (VIOFilter.cpp:421-427; called by removeOldLandmarks :393-419 and removeOutliers :429-443) for the n_remove
 * slots remove_slots[], then addNewLandmarks (:345-391) into the n_add free slots add_slots[] (slots freed by this very call included):
 * p0 = add_bearings[k] * depth (the caller's median scene depth, :358-366), Q = identity, zero cross-covariances,
 * initialPointVariance I on the diagonal.  new_num_slots = slots in use afterwards (>= 1 + the highest active slot; eqf_tiled_num_landmarks
 * returns it).  The geometry in force (eqf_tiled_set_geometry) must cover max(old, new) slots -- the rank's blocks of the marked slots
 * are cleared in Sll through it; shrink the geometry AFTER the call.  Host arrays; synchronises.  EQF_ERR_INVALID (before any effect) if
 * a removed slot is empty, an added slot taken, an active slot would fall above new_num_slots, or the working set would GROW over a slot
 * that this call does not fill (every slot in [old count, new_num_slots) must be in add_slots: a slot that was never initialised has no
 * unit diagonal block to decouple it); EQF_ERR_CAPACITY beyond the capacity. */
int eqf_tiled_edit_landmarks(eqf_tiled* t, int n_remove, const int* remove_slots, int n_add, const int* add_slots,
    const double* add_bearings, double depth, int new_num_slots, double* Sll, int ldl);
/* First half of the update (VIOFilter.cpp:264-277, EqFMatrices.cpp:319-344, :221-235): residual, C0i, lift rows; then the operands of
 * the two factorisations from the local blocks:
 *   M (2 nlr x ldm): columns [0, 2 nlc) S_IJ = C_I Sigma_IJ C_J^T (+ measurementVariance on the global diagonal), [2 nlc, 5 nlc)
 *     (C Sigma)_IJ, [5 nlc, 5 nlc + 18) the narrow right-hand sides [(C Sigma)_Ib (11) | delta_I | V_I = C_I Z_I (6)];
 *   E (3 nlr x lde): columns [0, 3 nlc) Sigma_IJ - Pg_I^T Pg_J (Schur complement of Sigma_e = Sigma[6:, 6:] after its five base
 *     coordinates, EqFMatrices.cpp:239), [3 nlc, 3 nlc + 11) [Z_I (6) | -Pg_I^T Lg^-1 (5)];  G11 (11 x 11): the base part of
 *     [Zt | Et]^T [Zt | Et].
 * bearings[N][3] host memory, in SLOT order (an inactive slot's entry is ignored); copied into a pinned staging buffer before the call
 * returns -- the upload and everything else is enqueued, nothing waits for the device.  bearings = NULL: already staged by
 * eqf_tiled_stage_bearings (a caller that replays a captured hipGraph of the update stages, then launches the graph).
 * eqf_tiled_pingpong: bit 0 / bit 1 = which of the two state / base-panel buffers is current (the device pointers of an update's launches
 * depend on it: a captured graph is valid for one value). */
int eqf_tiled_stage_bearings(eqf_tiled* t, const double* bearings);
int eqf_tiled_pingpong(eqf_tiled* t);
int eqf_tiled_update_prep(eqf_tiled* t, const double* bearings, const double* Sll, int ldl, double* M, int ldm, double* E, int lde,
    double* G11);
/* Second half (VIOFilter.cpp:279-297, EqFMatrices.cpp:173-275): acc (18 x ldacc, columns = 3 N in GLOBAL landmark order) = sum_k
 * Yn_k^T Y_k, Gnn (18 x 18) = sum_k Yn_k^T Yn_k, G11 (11 x 11) complete; gamma = K delta, bundleLift, Delta, X <- Delta X, bias +=
 * gamma[0:6]; the base panel's share of Sigma - K C Sigma.  (The local blocks were downdated by eqf_tile_gemm_tn as the solved block
 * rows arrived.) */
int eqf_tiled_update_finish(eqf_tiled* t, const double* acc, int ldacc, const double* Gnn, const double* G11);

int eqf_tiled_synchronize(eqf_tiled* t);
int eqf_tiled_num_landmarks(eqf_tiled* t);
int eqf_tiled_get_time(eqf_tiled* t, double* time);
int eqf_tiled_device_error(eqf_tiled* t);
/* as eqf_get_state_estimate / eqf_get_origin / eqf_get_group / eqf_get_bias / eqf_get_last_update / eqf_get_integrator */
int eqf_tiled_get_state_estimate(eqf_tiled* t, double* pose_q, double* pose_x, double* velocity, double* p);
int eqf_tiled_get_origin(eqf_tiled* t, double* pose_q, double* pose_x, double* velocity, double* p);
int eqf_tiled_get_group(eqf_tiled* t, double* A_q, double* A_x, double* w, double* Q_q, double* Q_a);
int eqf_tiled_get_bias(eqf_tiled* t, double* bias6);
int eqf_tiled_get_last_update(eqf_tiled* t, double* delta, double* gamma, double* Gamma);
int eqf_tiled_get_integrator(eqf_tiled* t, double* currentVelocity6, double* accumulatedVelocity6, double* accumulatedTime, int* initialised);
/* The replicated base rows Sigma[0:11, 0:11+3N] in the reference's index map (11 x ld, host memory). */
int eqf_tiled_get_base(eqf_tiled* t, double* dst, int ld);
/* State injection (as eqf_set_state; sigma_base = the first 11 rows of Sigma, 11 x ld, reference index map).  The caller fills Sll. */
int eqf_tiled_set_state(eqf_tiled* t, int N, const double* pose_q, const double* pose_x, const double* velocity, const double* p0,
    const double* A_q, const double* A_x, const double* w, const double* Q_q, const double* Q_a, const double* bias6,
    const double* sigma_base, int ld, double currentTime, const double* currentVelocity6, const double* accumulatedVelocity6,
    double accumulatedTime, int initialised);

/* ================================================================================================================================
 * The HOST LOOP of the partitioned filter behind the C ABI (csrc/eqf_tiledf.hip; round 5 -- until then it was Python): one eqf_tf per
 * rank of a Pr x Pc process grid (Pr | Pc; rank = pr * Pc + pc) IS VIOFilter (VIOFilter.h:41-88) for one filter whose Sigma is
 * partitioned over the grid.  Every rank makes the same calls with the same arguments.  The handle owns its eqf_tiled, its local matrix,
 * every exchange buffer and four HIP streams (two pairs with disjoint CU sets: reserve_cus CUs for the look-ahead factorisations, < 0 =
 * default 24 / EQF_TILED_RESERVE_CUS, 0 = plain streams).  What the ranks exchange goes through ONE callback:
 *   bcast(ctx, group, chain, root, buf, bytes, stream): broadcast `bytes` bytes of DEVICE memory at `buf` from `root` to the other members
 *   of `group` -- 0: my process row (root = process COLUMN of the sender), 1: my process column (root = process ROW), 2: every rank (root =
 *   rank) -- ordered on the HIP stream `stream` (the transfer may start when the stream reaches it; later work on the stream sees the data).
 *   chain = 0 / 1: the two factorisations of an update run side by side on different streams -- a collective library that cannot have two
 *   operations of one communicator in flight gets one communicator per chain.  Returns 0 on success.  With RCCL: ncclBroadcast(buf, buf,
 *   bytes, ncclChar, root, rowComm / colComm / worldComm [chain], stream).  A 1 x 1 grid needs no callback (comm = NULL).
 * eqf_tf_process_imu / eqf_tf_process_vision return EQF_OK, the EQF_SKIPPED_* code of the reference's silent early-outs, or an error
 * (EQF_ERR_UNSORTED: ids not strictly ascending, VIOFilter.cpp:239-240; EQF_ERR_CAPACITY; EQF_ERR_NUMERIC: a pivot of S or Sigma_e not
 * positive, looked at every check_every-th update).  Getters answer in the REFERENCE's landmark order (insertion order, :211-230); the
 * landmark slots behind it (eqf_tiled, above) are internal.  eqf_tf_get_sigma is collective (every rank calls it).
 * Options (eqf_tf_set_option): "lookahead" (1), "overlap_chains" (1 on one rank, 0 on a grid: the interleaved chains are not validated over RCCL on a node; EQF_TILED_OVERLAP_CHAINS), "burst" (1: IMU calls queued and sent as
 * bursts), "check_every" (1), "downdate_slices" (0: the covariance downdate Sigma - Y^T Y on the fp64 matrix cores, parity grade; 5 / 6 / 7: on the INTEGER matrix pipe from
 * that many 7-bit slices of Y's columns with exact accumulation -- Sigma within 1e-4 of the fp64 path from 6 slices on (measured 2e-6 at N = 200 .. 6e-5 at N = 4000), two thirds of the downdate's time;
 * round 6, csrc/eqf_tile.hpp), "chain_slices" (0: the two factorisations' trailing products on the fp64 matrix cores; 5 / 6 / 7: on the integer pipe in the same
 * way -- they forgive more than the downdate: five slices keep Sigma to 1e-8 and the pose to 3e-9 of the fp64 path on the bench stream), "downdate_early" (50: with the chains side by side and the fp64 downdate, the shares Y_k^T Y_k of this percentage of the S-chain's block rows are subtracted on the
 * E-chain's stream as soon as each block row is solved, the rest in one product behind the S-chain; 0: the whole downdate behind the S-chain, as until round 6),
 * "trsm_leaf" (0: a block row's triangular solve is split once into two solves and a product; > 0: recursively, down to this many 64-row blocks -- measured slower at N = 4000),
 * "profiling" (0: event brackets per phase, eqf_tf_get_phases in eqf_vio_amd_debug.h), "graphs" (0: hipGraph replay of an update on a
 * one-rank grid, see eqf_tf_graph_launches in eqf_vio_amd_debug.h).
 * ================================================================================================================================ */
typedef struct eqf_tf eqf_tf; /* opaque */
typedef struct eqf_tf_comm {
    void* ctx;
    int (*bcast)(void* ctx, int group, int chain, int root, void* buf, size_t bytes, void* stream);
} eqf_tf_comm;
int eqf_tf_create(const eqf_settings* settings, int capacity_landmarks, int block_landmarks, int Pr, int Pc, int rank, int device,
    int reserve_cus, const eqf_tf_comm* comm, eqf_tf** out);
void eqf_tf_destroy(eqf_tf* f);
int eqf_tf_set_option(eqf_tf* f, const char* name, int value);
/* VIOFilter::processIMUData (VIOFilter.cpp:120-131) / processVisionData (:232-302; ids[n] strictly ascending, bearings[n][3], host) */
int eqf_tf_process_imu(eqf_tf* f, double stamp, const double* omega, const double* accel);
int eqf_tf_process_vision(eqf_tf* f, double stamp, int n, const int* ids, const double* bearings);
int eqf_tf_synchronize(eqf_tf* f);
int eqf_tf_check(eqf_tf* f);        /* EQF_ERR_NUMERIC if a pivot was not positive since the last look (synchronises) */
int eqf_tf_device_error(eqf_tf* f); /* as eqf_device_error */
int eqf_tf_num_landmarks(eqf_tf* f);
int eqf_tf_num_slots(eqf_tf* f);
int eqf_tf_get_ids(eqf_tf* f, int* ids, int* slots); /* reference order; slots may be NULL */
int eqf_tf_get_time(eqf_tf* f, double* time);
int eqf_tf_get_state_estimate(eqf_tf* f, double* pose_q, double* pose_x, double* velocity, double* p);
int eqf_tf_get_bias(eqf_tf* f, double* bias6);
int eqf_tf_get_last_update(eqf_tf* f, double* delta, double* gamma, double* Gamma);
/* VIOFilter::stateCovariance (:306-309): dst (n x n, ld), n = 11 + 3 N in the reference's order (slot_order = 0) or 11 + 3 * slots over
 * all slots in use, holes included (slot_order = 1: tests of the hole invariants).  Host memory. */
int eqf_tf_get_sigma(eqf_tf* f, double* dst, int ld, int slot_order);
/* restart from a snapshot, as eqf_set_state (sigma: the DENSE covariance, every rank holds it once) */
int eqf_tf_set_state(eqf_tf* f, int N, const int* ids, const double* pose_q, const double* pose_x, const double* velocity, const double* p0,
    const double* A_q, const double* A_x, const double* w, const double* Q_q, const double* Q_a, const double* bias6, const double* sigma, int ld,
    double currentTime, const double* currentVelocity6, const double* accumulatedVelocity6, double accumulatedTime, int initialised);
int eqf_tf_get_churn_stats(eqf_tf* f, long long* stats3); /* removed_old, removed_outliers, added */
int eqf_tf_local_matrix(eqf_tf* f, double** ptr, int* rows, int* cols, int* ld);
const char* eqf_tf_last_error(eqf_tf* f);
const char* eqf_version(void);
/* "src_sha256=<hex>": sha256 over the library's sources (every .hip and .hpp file of csrc/, the two headers of include/, csrc/Makefile; concatenated in sorted
 * order) at build time -- lets a caller prove that the .so it loaded was built from the sources beside it (bench.py's `build` block). */
const char* eqf_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* EQF_VIO_AMD_H */
