/* pnp_hip.h - C ABI of libpnp_hip.so, the MI355X (gfx950) batched PnP pose solver.
 *
 * The reference (Benson516/pnp_solver_test) is pure Python and has no FFI of its own; the
 * boundary this library sits behind is the Python method PNP_SOLVER.solve_pnp
 * (scripts/PNP_SOLVER_LIB.py:144-203) and the per-pattern solvers it dispatches to.  Each
 * entry point below names the reference function whose results it reproduces.
 *
 * Conventions shared by all entry points
 *   - Arrays are dense, row-major, caller-allocated.  uv is (B,N,2) float32 pixel
 *     coordinates; xyz is the 3-D pattern, either one (N,3) float32 array shared by all
 *     problems (xyz_problem_stride == 0) or one per problem (xyz_problem_stride = number of
 *     floats between consecutive problems, normally 3*N).
 *   - K is the 3x3 camera matrix in float64, row-major (np_K_camera_est of the reference
 *     constructor, PNP_SOLVER_LIB.py:10-14).  It is inverted in float64 on the host, as the
 *     reference does with numpy.linalg.inv (:2811, :3097).
 *   - Arithmetic is always float64.  The _f32 entry points read float32 arrays and round
 *     their outputs to float32 once; the _f64 entry points take and return float64 arrays
 *     (the reference's own dtype) and otherwise behave identically: same kernels, same
 *     arguments, twice the bytes per problem.
 *   - ptr_kind: PNP_PTR_HOST = every array pointer is host memory (the library stages
 *     through device buffers and synchronises before returning); PNP_PTR_DEVICE = every
 *     array pointer is device memory on the current HIP device (the call only enqueues work
 *     on hip_stream and returns; nothing is allocated).
 *   - hip_stream: a hipStream_t cast to void*, or NULL for the default stream.
 *   - Return value: 0 on success, a negative PNP_E_* code otherwise; pnp_last_error() then
 *     returns a thread-local, human-readable description.  Numerically degenerate problems
 *     are not errors: as in the reference they produce NaN/inf in their own outputs.
 *   - Output pointers documented as optional may be NULL.
 *   - Thread safety: calls are independent as long as they do not share output buffers;
 *     host-pointer calls on one device serialise on that device's staging buffers.
 *   - Host-pointer calls never copy to device memory when they can help it: the kernels are
 *     launched on host memory and read and write it over the bus.  A call all of whose arrays
 *     are page-locked, mapped host memory (hipHostMalloc and the like) is one launch on those
 *     arrays.  Otherwise calls that move 8 MB or more are cut into chunks of up to 1 MB which
 *     twelve host threads per device carry through page-locked buffers and streams of the
 *     library's own: copy a chunk of the pageable arrays in, launch on the buffer, copy the
 *     results out, the threads overlapping each other; a side (inputs, outputs) that is
 *     page-locked is used as it is.  The threads are started by the first such call on a device
 *     (none exists before it) and stay, waiting, until the library is unloaded; concurrent
 *     callers share them chunk by chunk.  Results are those of one launch over the whole batch;
 *     work already queued on hip_stream is waited for.
 *   - The library reads no environment variable.
 */
#ifndef PNP_HIP_H
#define PNP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNP_PTR_HOST 0
#define PNP_PTR_DEVICE 1

#define PNP_OK 0
#define PNP_E_ARG (-1)     /* bad argument (NULL pointer, N out of range, ...) */
#define PNP_E_HIP (-2)     /* a HIP runtime call failed                         */
#define PNP_E_NODEVICE (-3)
#define PNP_E_RCCL (-4)    /* RCCL could not be opened, or one of its calls failed */

/* Tunables of the quaternion information-filter solver; the defaults are the constants
 * hard-coded in solve_pnp_QEIF_single_pattern (PNP_SOLVER_LIB.py:2791-2794, :2836,
 * :2845-2846, :2857, :2863, :2952). */
typedef struct pnp_qeif_params {
    double q_inv;     /* 1/Q with Q = 3^2/f^2, f = 225.68 (not taken from K)  -> 5659.05 */
    double r_quat;    /* process noise added to the quaternion block each step -> 0.1   */
    double r_delta;   /* process noise added to delta_1, delta_2               -> 0.01  */
    double omega0;    /* initial information matrix = omega0 * I               -> 1e-5  */
                      /* q_inv, r_quat, r_delta and omega0 must be > 0 (PNP_E_ARG otherwise) */
    int32_t max_it;   /* iteration cap                                         -> 14    */
    double rel_tol;   /* stop when |res - res_old| / res_old < rel_tol         -> 1e-2  */
    double res_old0;  /* initial res_old                                       -> 1e-7  */
} pnp_qeif_params;

/* Fills *p with the reference defaults listed above. */
void pnp_qeif_default_params(pnp_qeif_params *p);

/* Number of HIP devices visible to this process (0 if none, negative on error). */
int pnp_device_count(void);

/* Makes `device` the calling thread's current HIP device and prepares its context.
 * Optional: every entry point initialises the current device on first use. */
int pnp_init(int device);

/* Replaces solve_pnp_QEIF_single_pattern (PNP_SOLVER_LIB.py:2771-3025) with its model
 * QEKF_get_hx_H (:3902-3983) and reconstruction QEKF_reconstruct_R_t_m1 (:3542-3609),
 * for B independent problems of N points each.
 *   R_out   (B,9)  rotation, row-major                      required
 *   t_out   (B,3)  translation; t_out[2] is t3_est          required
 *   res_out (B)    res_norm at the last linearisation point optional
 *   iters_out (B)  iterations executed (1..max_it)          optional
 *   rpy_out (B,3)  roll, yaw, pitch in degrees, in the order of the reference's return
 *                  tuple (get_Euler_from_rotation_matrix, :4474-4517)   optional
 *   p: NULL = reference defaults. */
int pnp_qeif_batch_f32(const float *uv, const float *xyz, int64_t xyz_problem_stride,
                       const double K[9], int64_t B, int32_t N, const pnp_qeif_params *p,
                       float *R_out, float *t_out, float *res_out, int32_t *iters_out,
                       float *rpy_out, int32_t ptr_kind, void *hip_stream);

/* Replaces solve_pnp_single_pattern (PNP_SOLVER_LIB.py:205-430): num_it rounds (reference:
 * 3) of the linear system of get_Delta_A_all / get_B_all (:3064-3111), the closed-form
 * reconstruct_R_t_block_reconstruction (:4158-4364) and update_phi_3_est_m2 (:4002-4017).
 * res_out is the residual norm of the last round (:372-380).  Outputs as above. */
int pnp_dlt_batch_f32(const float *uv, const float *xyz, int64_t xyz_problem_stride,
                      const double K[9], int64_t B, int32_t N, int32_t num_it, float *R_out,
                      float *t_out, float *res_out, float *rpy_out, int32_t ptr_kind,
                      void *hip_stream);

/* Replaces solve_pnp_LM_single_pattern (PNP_SOLVER_LIB.py:2567-2769), the 12-state
 * Levenberg-Marquardt solver, with its model EKF2_get_hx_H (:3718-3836, options
 * is_hc0_4_6_linear=False, is_hc1_linear=True) and reconstruction EKF2_reconstruct_R_t_m1
 * (:3500-3540): num_it iterations (reference: 14, no early exit) of
 * dx = (J^T J + lambda I)^-1 J^T (z - h) with lambda = 1e-5 in the reference.  res_out is the
 * reprojection part of the residual at the last linearisation point (:2681).  A wavefront
 * solves 64 problems while their (2N + 3N) elements each fit 48 KB of LDS (N <= 38 with float32
 * arrays and a pattern per problem), and half, a quarter, ... as many beyond that, at the same
 * cost per wavefront; every N up to pnp_max_points() is accepted. */
int pnp_lm12_batch_f32(const float *uv, const float *xyz, int64_t xyz_problem_stride,
                       const double K[9], int64_t B, int32_t N, double lambda, int32_t num_it,
                       float *R_out, float *t_out, float *res_out, float *rpy_out,
                       int32_t ptr_kind, void *hip_stream);

/* pnp_qeif_batch_f32 for host pointers (PNP_PTR_HOST), allocating nothing.  It has no
 * counterpart in the reference; it is for callers that cannot allocate device memory: tight
 * serving loops, harnesses that forbid hipMalloc, code that must not change the process's memory
 * footprint.  pnp_qeif_batch_f32 passes pageable host arrays through page-locked buffers it
 * obtains from hipHostMalloc (and grows with hipHostFree + hipHostMalloc); this entry point copies
 * them to and from a static device array of PNP_NOALLOC_SCRATCH_BYTES bytes that every GPU the library runs on holds for
 * the life of the process.  It calls no hipMalloc, hipFree or hipHostMalloc.
 *   - Arguments and results are exactly those of pnp_qeif_batch_f32 with ptr_kind = PNP_PTR_HOST
 *     (bit-identical outputs); xyz_problem_stride is 0 or 3*N.  The kernels work on the static
 *     device array, so six-point problems may run the kernel whose lanes read their own rows
 *     where the host-pointer path, whose kernels read host memory, runs the LDS-staged one: the
 *     two return the same bits.
 *   - Limit: the staged arrays - B*(8N [+ 12N with per-problem xyz] + 68) bytes plus the shared
 *     pattern, each of eight pieces padded to 256 bytes - must fit the static array; a larger
 *     call fails with PNP_E_ARG and names the largest B that fits (about 89 000 six-point
 *     problems with per-problem xyz).  Split larger batches.
 *   - Every call on one GPU uses that GPU's one array.  Calls of one process on one GPU are
 *     serialised by a lock held until the results are back; nothing in the array has to start
 *     at zero and nothing is kept between calls.  The device address of the array is resolved
 *     by pnp_init(device) or, failing that, by the first call on a device. */
#define PNP_NOALLOC_SCRATCH_BYTES (16u << 20)
int pnp_qeif_batch_noalloc_f32(const float *uv, const float *xyz, int64_t xyz_problem_stride,
                               const double K[9], int64_t B, int32_t N, const pnp_qeif_params *p,
                               float *R_out, float *t_out, float *res_out, int32_t *iters_out,
                               float *rpy_out, void *hip_stream);

/* Replaces get_Euler_from_rotation_matrix(R, is_degree=True) (PNP_SOLVER_LIB.py:4474-4517)
 * for B matrices: R (B,9) -> rpy (B,3) = roll, yaw, pitch in degrees. */
int pnp_euler_deg_f32(const float *R, int64_t B, float *rpy, int32_t ptr_kind, void *hip_stream);

/* Replaces the pattern loop of solve_pnp (PNP_SOLVER_LIB.py:166-199): runs the QEIF solver
 * against n_patterns shared patterns, xyz (n_patterns,N,3), and keeps for every problem the
 * first pattern with the strictly smallest res_norm.  pattern_id_out (B) int32 is optional;
 * the other outputs are those of pnp_qeif_batch_f32. */
int pnp_qeif_multi_batch_f32(const float *uv, const float *xyz, int32_t n_patterns,
                             const double K[9], int64_t B, int32_t N, const pnp_qeif_params *p,
                             float *R_out, float *t_out, float *res_out, int32_t *iters_out,
                             float *rpy_out, int32_t *pattern_id_out, int32_t ptr_kind,
                             void *hip_stream);

/* float64 flavours of the five entry points above: every float* becomes double* (strides are
 * counted in doubles), everything else is unchanged.  They reproduce the reference to the
 * round-off of float64 rather than to that of float32 inputs. */
int pnp_qeif_batch_f64(const double *uv, const double *xyz, int64_t xyz_problem_stride,
                       const double K[9], int64_t B, int32_t N, const pnp_qeif_params *p,
                       double *R_out, double *t_out, double *res_out, int32_t *iters_out,
                       double *rpy_out, int32_t ptr_kind, void *hip_stream);
int pnp_dlt_batch_f64(const double *uv, const double *xyz, int64_t xyz_problem_stride,
                      const double K[9], int64_t B, int32_t N, int32_t num_it, double *R_out,
                      double *t_out, double *res_out, double *rpy_out, int32_t ptr_kind,
                      void *hip_stream);
int pnp_lm12_batch_f64(const double *uv, const double *xyz, int64_t xyz_problem_stride,
                       const double K[9], int64_t B, int32_t N, double lambda, int32_t num_it,
                       double *R_out, double *t_out, double *res_out, double *rpy_out,
                       int32_t ptr_kind, void *hip_stream);
int pnp_euler_deg_f64(const double *R, int64_t B, double *rpy, int32_t ptr_kind, void *hip_stream);
int pnp_qeif_multi_batch_f64(const double *uv, const double *xyz, int32_t n_patterns,
                             const double K[9], int64_t B, int32_t N, const pnp_qeif_params *p,
                             double *R_out, double *t_out, double *res_out, int32_t *iters_out,
                             double *rpy_out, int32_t *pattern_id_out, int32_t ptr_kind,
                             void *hip_stream);

/* Synthetic-input generator of the reference's drivers for B poses: replaces
 * get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True) (PNP_SOLVER_LIB.py:4442-4472)
 * followed by perspective_projection / perspective_projection_golden_landmarks
 * (:4532-4557, :4586-4621).  All inputs are float64, as the reference computes them:
 *   rpy_deg (B,3) roll, yaw, pitch in degrees;  t (B,3);  xyz (N,3) shared (stride 0) or
 *   (B,N,3) with xyz_problem_stride doubles between problems;  K (3,3) host pointer.
 *   quantize_q > 0 rounds the pixels half-to-even onto a grid of that step (is_quantized=True),
 *   quantize_q = 0 leaves them unquantised.   uv_out (B,N,2): float32 if uv_is_f64 == 0,
 *   float64 otherwise, ready for the solver entry points of the same flavour. */
int pnp_project_batch_f64(const double *rpy_deg, const double *t, const double *xyz,
                          int64_t xyz_problem_stride, const double K[9], int64_t B, int32_t N,
                          double quantize_q, void *uv_out, int32_t uv_is_f64, int32_t ptr_kind,
                          void *hip_stream);

/* Comparison epilogue of the reference's stress driver for B solved samples
 * (scripts/random_stress_test.py:362-396): replaces check_if_the_sample_passed
 * (scripts/TEST_TOOLBOX.py:55-62) and the landmark error sets of
 * compare_result_and_generate_result_dict / cal_LM_error_distances (TEST_TOOLBOX.py:252-286,
 * :312-343, :453-463), with get_Euler_from_rotation_matrix of the estimate fused in.
 *   R_est (B,9), t_est (B,3)   the solver's outputs (float32 or float64 flavour)
 *   uv_obs (B,M,2)             the pixels that were solved, for all M pattern points; NULL = the
 *                              ground-truth projections stand in (TEST_TOOLBOX.py:317-319)
 *   rpy_gt_deg (B,3) float64   ground-truth roll, yaw, pitch in degrees
 *   distance_gt_cm (B) float64 ground-truth depth in cm (data_id_dict['distance'], :281)
 *   xyz (M,3) float64          the golden pattern, shared by all samples
 *   bound[4] (host)            pass bounds for depth[cm], roll, pitch, yaw; NULL = 10, 10, 10, 10
 *   normalize                  != 0 scales the landmark errors by the true depth in metres
 *                              (the *_normalize fields), 0 leaves them in pixels
 * Outputs, each optional but at least one:
 *   drpy_err (B,4)   estimate - truth for depth[cm], roll, pitch, yaw
 *   pass_mask (B)    bit k set when |drpy_err[k]| < bound[k]
 *   lm_err (B,6)     average and maximum landmark distance for observed-vs-truth,
 *                    reprojection-vs-observed, reprojection-vs-truth, in that order
 *   lm_key (B,3)     index of the pattern point with the (first) maximum, -1 if all are 0
 * The ground-truth translation is the ESTIMATED direction scaled to the true depth, as the
 * driver does (:367-368). */
int pnp_compare_batch_f32(const float *R_est, const float *t_est, const float *uv_obs,
                          const double *rpy_gt_deg, const double *distance_gt_cm, const double *xyz,
                          const double K[9], int64_t B, int32_t M, const double bound[4],
                          int32_t normalize, float *drpy_err, int32_t *pass_mask, float *lm_err,
                          int32_t *lm_key, int32_t ptr_kind, void *hip_stream);
int pnp_compare_batch_f64(const double *R_est, const double *t_est, const double *uv_obs,
                          const double *rpy_gt_deg, const double *distance_gt_cm, const double *xyz,
                          const double K[9], int64_t B, int32_t M, const double bound[4],
                          int32_t normalize, double *drpy_err, int32_t *pass_mask, double *lm_err,
                          int32_t *lm_key, int32_t ptr_kind, void *hip_stream);

/* Optional exchange of result arrays between the ranks of a sharded solve (one process per GPU,
 * every rank solving its own contiguous slice of the batch; the reference is single-process and
 * has no counterpart).  The solve itself needs no communication; these four entry points are for
 * a consumer that wants every rank's poses in one place.  They run on RCCL (xGMI between the GPUs
 * of a node), which is opened at run time by the first of them that is called: the library loads
 * without RCCL, and then these return PNP_E_RCCL.
 *
 * pnp_comm_unique_id  fills id_out[PNP_COMM_ID_BYTES] on ONE rank (ncclGetUniqueId); the caller
 *                     hands the bytes to every other rank by means of its own (a file, MPI, a
 *                     torch.distributed broadcast).
 * pnp_comm_init       every rank, with its device current: joins the communicator of `world`
 *                     ranks (ncclCommInitRank, collective) and returns it in *comm_out.
 * pnp_comm_destroy    releases it.
 * pnp_gather_results  send: this rank's bytes_per_rank bytes; recv: world * bytes_per_rank bytes,
 *                     the ranks' arrays in rank order, on every rank (root < 0: ncclAllGather) or
 *                     on rank `root` only (root >= 0: grouped ncclSend / ncclRecv, so that
 *                     world - 1 slices arrive over world - 1 different links; recv may be NULL on
 *                     the other ranks).  Device pointers; the same bytes_per_rank on every rank
 *                     (pad shorter slices); enqueued on hip_stream, not synchronised.
 *                     Collective: every rank of the communicator must call it. */
#define PNP_COMM_ID_BYTES 128
int pnp_comm_unique_id(void *id_out);
int pnp_comm_init(const void *id, int32_t rank, int32_t world, void **comm_out);
int pnp_comm_destroy(void *comm);
int pnp_gather_results(const void *send, void *recv, int64_t bytes_per_rank, int32_t root,
                       int32_t rank, int32_t world, void *comm, void *hip_stream);

/* How many lanes of a wavefront cooperate on one problem: 1, 2, 4, 8, 16, 32 or 64
 * (64 = one wavefront per problem); 0 restores the automatic choice from (B, N).  The automatic
 * choice keeps a workgroup's staged points within 48 KB of LDS; a forced count may go down to
 * what 160 KB hold (one workgroup per compute unit).
 * Belongs to the calling thread: it changes that thread's launches only (a host-pointer call
 * carries it to the library's own threads).  Meant for tuning and tests.  No reference
 * counterpart. */
int pnp_set_lanes_per_problem(int32_t lanes);

/* Which arithmetic form the QEIF entry points run.  Both follow the same recurrence and agree
 * to float64 round-off; they differ in cost.
 *   PNP_QEIF_FORM_POINTS   every iteration visits every point (78 N fused operations)
 *   PNP_QEIF_FORM_MOMENTS  50 sums over the points are formed once, an iteration then costs
 *                          about 700 fused operations whatever N is; one lane per problem
 *   PNP_QEIF_FORM_AUTO     moments when N >= 10, B >= 2048, one pattern and no forced lane
 *                          count; points otherwise (the default)
 * Belongs to the calling thread, like pnp_set_lanes_per_problem.  No reference counterpart. */
#define PNP_QEIF_FORM_AUTO 0
#define PNP_QEIF_FORM_POINTS 1
#define PNP_QEIF_FORM_MOMENTS 2
int pnp_set_qeif_form(int32_t form);

/* The form (POINTS or MOMENTS) the calling thread's next single-pattern QEIF launch would use
 * for (B, N). */
int pnp_get_qeif_form(int64_t B, int32_t N);

/* The value the next float32 launch would use for (B, N). */
int pnp_get_lanes_per_problem(int64_t B, int32_t N, int64_t xyz_problem_stride);

/* The kernel the calling thread's last solver launch dispatched, by the name a profiler shows
 * for it without the argument list (e.g. "pnp::qeif_rows6_kernel<float>"), "" before the first
 * launch; for a chunked host-pointer call, the kernel of its chunks.  pnp_last_launch gives that
 * launch's geometry: workgroups, threads per workgroup, bytes of dynamic LDS (each pointer may
 * be NULL).  No reference counterpart: measurement harnesses use it to name what they timed. */
const char *pnp_last_kernel_name(void);
void pnp_last_launch(int64_t *grid, int32_t *block, int32_t *lds_bytes);

/* Number of host threads the library keeps for chunked host-pointer calls on `device`: 0 until
 * the first such call, 12 afterwards. */
int pnp_host_lane_threads(int device);

/* Largest N accepted by the solvers. */
int pnp_max_points(void);

/* Thread-local description of the last failure in the calling thread ("" if none). */
const char *pnp_last_error(void);

/* Library version string. */
const char *pnp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PNP_HIP_H */
