/* eqf_vio_amd -- test, measurement and developer hooks of the library.  NOT part of the drop-in boundary: include/eqf_vio_amd.h is what a
 * maintainer binding pvangoor/eqf_vio's VIOFilter (eqf_vio/include/eqf_vio/VIOFilter.h:64-88) sees; nothing here is needed to run a filter.
 * Everything below is exported by the same libeqf_vio_amd.so and used by tests/, bench.py's roofline leg, scripts/ and
 * tests/tiled_reference.py (the partitioned filter's schedule spelt out over the dense tile kernels).
 */
#ifndef EQF_VIO_AMD_DEBUG_H
#define EQF_VIO_AMD_DEBUG_H

#include "eqf_vio_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook for block-level parity (A0 / B / C0 blocks): the linearisation blocks of filter b as the most
 * recent single-step launch of the split propagate path (k_build_blocks: EQF_IMU_BURST=0, EQF_SPLIT_PROPAGATE=1) left
 * them, and the per-landmark output blocks.  common[31] = T, B[0:2,0:3] (6), B[2:5,0:3] (9), R_A = B[2:5,3:6] (9),
 * A0[2:5,0:2] (6), row-major (EqFMatrices.cpp:288-289, :364-368).  rec[N][27] per landmark: D = I + T A0[ii] (9),
 * Lw = -T B[5+3i.., 0:3] (9), Lv = T A0[5+3i.., 2:5] (9) (EqFMatrices.cpp:294-314, :370-380).  c0[N][6] = C0i, the 2 x 3
 * block of EqFOutputMatrixC (EqFMatrices.cpp:319-344).  Any pointer may be NULL.  fp64 handles only. */
int eqf_debug_get_blocks(eqf_filter* f, int b, double* common, double* rec, double* c0);
/* Fault injection for the in-launch hand-offs (tests only): from the next update on, the workgroup of k_chol_resident with role (kind:
 * 0 S-chain / 1 E-chain; role: 0 row head H(R), 1 interior tile T(R, C), 2 right-hand-side tile W(t = R, C)) leaves without doing or
 * publishing anything, as if it had never been scheduled: its consumers time out after 0.5 s, bit 128 is raised, the launch unwinds, the
 * covariance downdate does not run.  kind < 0 switches the injection off.  The handle needs eqf_reset / eqf_set_state afterwards. */
int eqf_debug_drop_role(eqf_filter* f, int kind, int role, int R, int C);
/* Developer toggles by name (tests and measurements; a production caller needs none of them).  EQF_ERR_INVALID: unknown name.
 *   "cs_in_burst"  1 (default): an IMU burst closed by a vision step also leaves the landmark columns of C Sigma and S = C Sigma C^T + R,
 *                  formed from the covariance blocks its workgroups hold in registers; the update's prep work then reads 12 columns of
 *                  Sigma per landmark instead of all of them -- at the throughput sizes (many filters, N >= 400), where it pays.
 *                  2: on every burst that runs as two launches.  0: the prep launch forms them.  Bit for bit the same either way.
 *   "device_edit"  1 (default): a vision frame's landmark bookkeeping -- the landmarks that left (VIOFilter.cpp:393-419), the outlier gate
 *                  (:429-443), the new landmarks (:345-391) -- is ONE launch that decides and acts on the device; the id lists of the handle
 *                  follow when the caller next touches it and no frame is redone.  0: separate launches, a frame with an outlier is redone
 *                  from the host.  Bit for bit the same either way.
 *   "burst_lm" / "burst_rows" / "ring_ahead2"   launch shapes of an IMU burst: landmarks per builder workgroup (0 = by launch size, 4, 8, 16), row
 *                  landmarks per wavefront of the block kernel (0 = by launch size, 1, 2, 4), and whether the two-row block kernel requests a
 *                  step's constants two steps ahead (1) or one (0, default: two measured slower).  Bit for bit the same results whatever is chosen.
 *   "res_tickets"  a workgroup of the update launch draws its place in its filter's dependency order from a counter when it starts (no assumption
 *                  about the hardware's dispatch order; a kernel build of its own) instead of reading it off its block index (rounds 3-5: relies
 *                  on in-order dispatch, guarded by the time-outs).  0 (default): never.  1: on grids of at least six times the resident slots (16+
 *                  filters of N = 200, N >= ~700: +1-2 % per update there).  2: on every grid larger than the chip with a prep launch in front
 *                  (below six times the slots +3 .. 15 us per update).  Same results.
 *   "burst_fused_max_x10"    the one-launch IMU burst (k_burst_fused) is used up to value / 10 workgroups per CU (default: 1.25).
 *   "e_sigma_min_percu_x10"  the two-per-CU build of the update launch reads the E-chain's tiles in Sigma itself from value / 10 chain roles per
 *                  CU on (default 2.4: every such grid); below, the prep launch copies Sigma[6:, 6:].  Launch shapes only: same results. */
int eqf_debug_option(eqf_filter* f, const char* name, int value);
/* The launch shape of the handle's most recent IMU burst that a vision step closed (bench.py prices the kernels by it): shape8[0] landmarks per builder workgroup,
 * [1] row landmarks per wavefront of the block kernel, [2] 1 = one launch (k_burst_fused), [3] 1 = the burst also left the landmark columns
 * of C Sigma and S (the update's prep work does not read Sigma then), [4] builder workgroups per filter, [5] block-kernel workgroups per
 * filter, [6] steps in the burst, [7] reserved.  Does not touch the device or flush anything. */
int eqf_debug_launch_shape(eqf_filter* f, int* shape8);

/* Per-kernel-class timing with HIP events on the handle's stream (bench.py roofline leg).
 * eqf_profile_get: for class c in [0, EQF_PROF_CLASSES) -> launches and total milliseconds.  The total is, per launch shape
 * within the class (chain step index, burst length), the median bracket times the number of launches of that shape, minus
 * the calibrated cost of an empty bracket: an event bracket also contains the time the stream waited for the host. */
#define EQF_PROF_PROPAGATE 0
#define EQF_PROF_UPDATE_PREP 1
#define EQF_PROF_CHOL_STEP 2
#define EQF_PROF_REDUCE 3
#define EQF_PROF_FINISH 4
#define EQF_PROF_DOWNDATE 5
#define EQF_PROF_CHURN 6
#define EQF_PROF_DENSE 7 /* k_dense_build + the two k_dense_gemm launches of the dense Riccati backend */
#define EQF_PROF_BURST 8 /* k_burst_build + k_burst_riccati: one bracket per burst of integrateUpToTime steps */
#define EQF_PROF_CHOL_DD 9 /* the one k_chol_step64 launch per update that also carries Sigma - Y^T Y (64-wide path) */
#define EQF_PROF_CHOL_RESIDENT 10 /* k_chol_resident: the whole factorisation part of an update as one launch */
#define EQF_PROF_CLASSES 11
int eqf_profile_enable(eqf_filter* f, int on);
int eqf_profile_get(eqf_filter* f, int cls, long long* launches, double* total_ms);
const char* eqf_profile_class_name(int cls);

/* ---- the partitioned filter's host loop (eqf_tf_*): phase timers, the rank's eqf_tiled, hipGraph replay counter */
int eqf_tf_get_phases(eqf_tf* f, double* ms7);
const char* eqf_tf_phase_name(int i);
void* eqf_tf_tiled_handle(eqf_tf* f); /* the rank's eqf_tiled (getters of the replicated state in SLOT order, tests) */
/* One rank: the launch sequence of an update (~2000 launches at N = 4000) is captured once per (slots in use, buffer parity) as a hipGraph
 * and replayed with one hipGraphLaunch -- option "graphs"; OFF by default: on ROCm 7.2 the replay takes the GPU 1.6 x
 * (N = 4000) to 4 x (N = 1000) as long as the plain launches on four streams (csrc/eqf_tiledf.hip).  Updates replayed from a graph so far: */
long long eqf_tf_graph_launches(eqf_tf* f);

/* ---- Dense tile kernels of the distributed factorisations, on CALLER-OWNED device memory of HIP device `device`, enqueued on
 * `stream` (a hipStream_t, NULL = the default stream) without synchronising; the caller's current device is restored.
 * eqf_tile_gemm_tn: C (m x n, ldc) += alpha A^T B for A (k x m, lda), B (k x n, ldb), row-major -- every O(n^3) product of the
 *   distributed update in the block-ROW form of the factorisation: trailing updates U_ki^T U_kj and right-hand sides U_ki^T Y_kt
 *   (VIOFilter.cpp:276-277, EqFMatrices.cpp:239), the downdate Sigma_IJ -= Y_kI^T Y_kJ (VIOFilter.cpp:297), the reductions.
 *   mask_rb > 0: C is the matrix part of a block-cyclic local matrix whose strictly-lower blocks are never read; tiles entirely below
 *   the block diagonal are skipped (row r is in global block (rblk0 + r / mask_rb) * Pr + pr, column c in (cblk0 + c / mask_cb) * Pc + pc).
 *   The epilogue is C += alpha * acc as fire-and-forget global_atomic_add_f64 (one writer per element and launch: deterministic).  Two
 *   consequences for a caller: C must be ordinary (coarse-grained) device memory -- hipMalloc / a torch CUDA tensor; on fine-grained or
 *   host-coherent allocations hardware fp64 atomics may be unsupported -- and for alpha other than +-1 the result is rounded twice
 *   (alpha * acc, then the addition) instead of once as fma(alpha, acc, C); every product of the filter uses alpha = +-1.
 * eqf_tile_downdate = eqf_tile_gemm_tn with alpha = -1 and no mask.
 * eqf_tile_potrf: A (n x n, ld, lower triangle) <- L with A = L L^T: the diagonal block of a block row; drec [ceil(n / 64)][5120]
 *   receives, per 64-wide block column, L_jj and the inverses of its four 16 x 16 diagonal blocks (what eqf_tile_trsm multiplies with);
 *   info (device int, may be NULL) is or-ed with 1 if a pivot is not positive.
 * eqf_tile_trsm: right = 1: B (m x n, ldb) <- B L^-T; right = 0: B (n x m, ldb) <- L^-1 B (the solved block row [U_k,k+1.. | Y_k]).
 * eqf_tile_propagate: one structured Riccati step of a (3 nI x 3 nJ) tile from explicit block arrays (kept for the block-level tests;
 *   the closed loop uses eqf_tiled_propagate):
 *     out = (D_I in + L_I Sigma_bJ) D_J^T + (L_I Sigma_bb + D_I Sigma_Ib) L_J^T + T (B_I R B_J^T) [+ diag_noise I on a diagonal tile] */
/* A HIP stream whose kernels only run on the CUs [first_cu, first_cu + num_cus) (complement = 0) or on all the others (complement = 1)
 * (hipExtStreamCreateWithCUMask).  The look-ahead of the distributed factorisations factors the next diagonal block -- one workgroup
 * with 119 KB of LDS -- on a few reserved CUs while the trailing update fills the rest of the chip; without the reservation the
 * update's workgroups (two per CU, 147 KB of LDS) never leave room for it.  eqf_stream_destroy waits for the stream and keeps a masked stream
 * for the next request of the same device and mask (a masked stream is a hardware queue of its own, and destroyed ones were measured not to
 * come back: see csrc/eqf_tiled.hip); any other stream is destroyed. */
int eqf_stream_create_masked(int device, int first_cu, int num_cus, int complement, void** out);
int eqf_stream_destroy(int device, void* stream);
int eqf_tile_gemm_tn(int device, void* stream, double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k,
    double alpha, int mask_rb, int mask_cb, int rblk0, int Pr, int pr, int cblk0, int Pc, int pc);
/* C (n x n, ldc; symmetric up to rounding, blocks of rb): C[r][c] <- C[c][r] wherever r / rb > c / rb.  Completes a block-upper-masked
 * eqf_tile_gemm_tn on a rank whose local matrix is symmetric (square process grid, diagonal rank): Sigma - K C Sigma = Sigma - Y^T Y
 * (VIOFilter.cpp:297) at half the flops there. */
int eqf_tile_mirror(int device, void* stream, double* C, int ldc, int n, int rb);
int eqf_tile_propagate(int device, void* stream, double* out, const double* in, int ld, int nI, int nJ, const double* D_I,
    const double* L_I, const double* D_J, const double* L_J, const double* Sbb, const double* SbI, int ldbI, const double* SbJ,
    int ldbJ, const double* BnI, const double* BnJ, const double* R6, double T, double diag_noise, int is_diag);
int eqf_tile_downdate(int device, void* stream, double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb,
    int k);
/* eqf_tile_downdate_i8 (round 6): C (m x n, ldc) -= A^T B as eqf_tile_downdate, but on the INTEGER matrix pipe: every column of A and B is scaled
 *   by a power of two and cut into `slices` (5, 6 or 7) signed 7-bit pieces, the slice pairs are multiplied on v_mfma_i32_32x32x32_i8 with exact
 *   int32 accumulation and recombined in fp64 (csrc/eqf_tile.hpp).  The only error is the truncation of an entry below 2^-(6 + 7 (slices - 1)) of
 *   its column's largest entry; k <= 70 000.  mask_rb > 0 (m == n): C is a symmetric local matrix in blocks of mask_rb, tiles entirely below the
 *   block diagonal are skipped (eqf_tile_mirror completes them).  workspace: caller-owned device memory of at least
 *   eqf_tile_i8_workspace_bytes(m, n, k, slices, A == B) bytes (the slices of both operands, once if they are the same matrix). */
size_t eqf_tile_i8_workspace_bytes(int m, int n, int k, int slices, int same_operand);
int eqf_tile_downdate_i8(int device, void* stream, double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k,
    int slices, int mask_rb, void* workspace, size_t workspace_bytes);
/* eqf_tile_gemm_tn_i8 (round 6): the same integer-pipe product behind eqf_tile_gemm_tn's block mask, alpha = -1: what a block row of the two
 *   factorisations of an update subtracts from its trailing rows (eqf_tf_set_option "chain_slices").  The mask (mask_rb > 0) covers the first
 *   mask_cols columns of C (the matrix part); the columns from mask_cols on (right-hand sides) are always formed.  When A is a column range of
 *   B that starts at a multiple of 32 (same rows of memory, lda == ldb) the operands are cut once.  workspace: at least
 *   eqf_tile_i8_workspace_bytes(m, n, k, slices, 0) bytes. */
int eqf_tile_gemm_tn_i8(int device, void* stream, double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k,
    int slices, int mask_rb, int mask_cb, int rblk0, int Pr, int pr, int cblk0, int Pc, int pc, int mask_cols, void* workspace,
    size_t workspace_bytes);
int eqf_tile_potrf(int device, void* stream, double* A, int ld, int n, double* drec, int* info);
int eqf_tile_trsm(int device, void* stream, const double* A, int ld, int n, const double* drec, double* B, int ldb, int m, int right);


#ifdef __cplusplus
}
#endif
#endif /* EQF_VIO_AMD_DEBUG_H */
