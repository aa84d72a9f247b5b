/*
 * rsparse_wrmf_hip.h -- C ABI of librsparse_wrmf_hip.so: the MI355X (gfx950) implementation of
 * rsparse's WRMF / ALS hot path.
 *
 * Plain C: pointers and sizes only, no R (SEXP), Rcpp, Armadillo or torch types.  Every entry
 * point returns an int status (0 = ok); nothing throws or exits across the boundary.  After a
 * non-zero status rsparse_hip_last_error() returns a description (thread-local).
 *
 * Two layers:
 *
 *  (1) stateless drop-ins -- the argument lists of the four generated `.Call` targets
 *      _rsparse_als_implicit_{double,float} / _rsparse_als_explicit_{double,float}
 *      (reference: src/RcppExports.cpp:329-415, src/wrmf_implicit.cpp:4-31,
 *      src/wrmf_explicit.cpp:4-27) with the S4 / arma objects flattened to the raw buffers those
 *      functions extract (src/utils.cpp:69-78 dgCMatrix slots, :115-128 float32 payload).
 *      Host pointers in, `Y` mutated in place, loss returned -- exactly the reference contract
 *      (R/model_WRMF.R:492,513 "Y is modified in-place").  They upload, run, download.
 *
 *  (2) a device-resident layer for callers that keep the matrices in HBM across half-iterations
 *      (the reference re-passes host memory every call; at 10M x 1M that would re-upload >4 GB of
 *      CSC per half-iteration).  Device pointers + a HIP stream; no implicit synchronisation
 *      unless a host result is requested.
 *
 * Layouts (identical to the reference):
 *   Conf    CSC: col_ptrs int32[n_cols+1], row_indices int32[nnz] 0-based, values (f64 on the
 *           host boundary, f32 once resident); one CSC column = one row solved.
 *   X       rank x n_rows  column-major (entity vectors contiguous), read only
 *   Y       rank x n_cols  column-major, in: CG warm start, out: solution
 *   XtX     rank x rank, must already contain + lambda*I (R/model_WRMF.R:474-486)
 *
 * Arithmetic: the *_float entry points and the device-resident layer (2) compute in fp32 (the
 * reference's precision="float" build; stated tolerance vs the reference CPU path: 1e-4 relative
 * Frobenius on the factor matrices).  The *_double entry points and the fp64 device layer (3)
 * compute in double, as als_implicit<double> / als_explicit<double> do (src/wrmf_implicit.cpp:5-14,
 * src/wrmf_explicit.cpp:5-14; precision = "double" is the R constructor's default, R/model_WRMF.R:82).
 */
#ifndef RSPARSE_WRMF_HIP_H
#define RSPARSE_WRMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define RSPARSE_HIP_OK 0
#define RSPARSE_HIP_ERR_INVALID 1     /* bad argument (NULL pointer, negative size, rank <= 0 ...) */
#define RSPARSE_HIP_ERR_UNSUPPORTED 2 /* variant not on the device path: caller keeps its CPU path   */
#define RSPARSE_HIP_ERR_RUNTIME 3     /* HIP runtime / out of memory / no device                    */
#define RSPARSE_HIP_ERR_NUMERIC 4     /* a per-row system was singular for the general solver too (the   */
                                      /* reference's arma::solve throws -> R error, RcppExports.cpp:374,392) */

/* solver codes: inst/include/wrmf.hpp:16-18 */
#define RSPARSE_SOLVER_CHOLESKY 0
#define RSPARSE_SOLVER_CONJUGATE_GRADIENT 1
#define RSPARSE_SOLVER_NNLS 2 /* sequential coordinate descent, inst/include/nnls.hpp:10-48 */

#define RSPARSE_HIP_MAX_RANK 256     /* fp32 entry points and layer (2); ranks 129..256 run on one generic kernel family */
#define RSPARSE_HIP_MAX_RANK_F64 128 /* the *_double entry points and layer (3) */

const char* rsparse_hip_last_error(void);
int rsparse_hip_abi_version(void);
/* number of visible HIP devices (0 if none / runtime unavailable) */
int rsparse_hip_device_count(void);
int rsparse_hip_set_device(int device);

/* ------------------------------------------------------------------------------------------------
 * (1) stateless drop-ins, host pointers
 * ---------------------------------------------------------------------------------------------- */

/* replaces als_implicit_float  (src/wrmf_implicit.cpp:17-31 -> als_implicit<float>,
 * inst/include/wrmf_implicit.hpp:90-305).  n_rows/n_cols/col_ptrs/row_indices/values are the
 * dgCMatrix slots Dim[0], Dim[1], p, i, x.  rank = nrow(X).  n_threads is accepted and ignored.
 * with_biases: Cholesky and NNLS only (XtX is then (rank-1) x (rank-1)); with conjugate_gradient -> ERR_UNSUPPORTED
 * (the reference cannot run that combination either, wrmf_implicit.hpp:189,197).
 * global_bias >= sqrt(epsilon of the element type) (wrmf_implicit.hpp:108-109; 3.45e-4 for _float, 1.49e-8 for _double;
 * smaller values count as zero): every solver -- Cholesky and NNLS with or without biases (:146-153, 228-229, 262-270),
 * conjugate gradient without biases (cg_solver_implicit_global_bias, :35-57, 203; marked "very poor numerical
 * precision" in the reference and restated as written: note that it solves  lhs y = X_nnz c + base - g X_nnz (c - 1),
 * the Cholesky branch  lhs y = X_nnz c + base).
 * global_bias_base (no biases only; may be NULL): the vector -global_bias * rowSums(X), `rank` entries (:111-112).
 * global_bias_base_len = the number of entries the caller's buffer holds.  The R driver allocates rank - 1
 * (R/model_WRMF.R:292) while the reference's C++ assigns and reads `rank` (a local re-allocation on the write, one
 * element past the R vector on the read); this library never touches more than the stated length: with
 * initialize_bias_base != 0 it writes min(len, rank) entries, with initialize_bias_base == 0 it reads the vector only
 * if len >= rank and otherwise recomputes it from X (its definition).
 * *loss_out = the value the reference returns (loss / nnz). */
int rsparse_hip_als_implicit_float(int n_rows, int n_cols, const int32_t* col_ptrs,
                                   const int32_t* row_indices, const double* values,
                                   const float* X, float* Y, const float* XtX, int rank,
                                   double lambda, int n_threads, unsigned solver,
                                   unsigned cg_steps, int with_biases, int is_x_bias_last_row,
                                   double global_bias, float* global_bias_base, int global_bias_base_len,
                                   int initialize_bias_base, double* loss_out);

/* replaces als_implicit_double (src/wrmf_implicit.cpp:5-14 -> als_implicit<double>).  Buffers are f64 like the
 * reference's and the device computes in f64 (layer 3 below): every solver, biases and the global bias as for _float. */
int rsparse_hip_als_implicit_double(int n_rows, int n_cols, const int32_t* col_ptrs,
                                    const int32_t* row_indices, const double* values,
                                    const double* X, double* Y, const double* XtX, int rank,
                                    double lambda, int n_threads, unsigned solver,
                                    unsigned cg_steps, int with_biases, int is_x_bias_last_row,
                                    double global_bias, double* global_bias_base, int global_bias_base_len,
                                    int initialize_bias_base, double* loss_out);

/* replaces als_explicit_float (src/wrmf_explicit.cpp:17-27 -> als_explicit<float>,
 * inst/include/wrmf_explicit.hpp:33-174).  cnt_X has n_rows entries (used only for the
 * dynamic_lambda regulariser term of the loss, :160-170); may be NULL when !dynamic_lambda. */
int rsparse_hip_als_explicit_float(int n_rows, int n_cols, const int32_t* col_ptrs,
                                   const int32_t* row_indices, const double* values,
                                   const float* X, float* Y, const float* cnt_X, int rank,
                                   double lambda, unsigned n_threads, unsigned solver,
                                   unsigned cg_steps, int dynamic_lambda, int with_biases,
                                   int is_x_bias_last_row, double* loss_out);

/* replaces als_explicit_double (src/wrmf_explicit.cpp:5-14 -> als_explicit<double>); f64 arithmetic on the device */
int rsparse_hip_als_explicit_double(int n_rows, int n_cols, const int32_t* col_ptrs,
                                    const int32_t* row_indices, const double* values,
                                    const double* X, double* Y, const double* cnt_X, int rank,
                                    double lambda, unsigned n_threads, unsigned solver,
                                    unsigned cg_steps, int dynamic_lambda, int with_biases,
                                    int is_x_bias_last_row, double* loss_out);

/* replaces the R-side Gramian  XtX = tcrossprod(X) + fl(diag(lambda))  (R/model_WRMF.R:474-486,
 * :347-353).  Host pointers; X is rank x n column-major; XtX_out rank x rank. */
int rsparse_hip_gramian_float(const float* X, int rank, int64_t n, double lambda, float* XtX_out);
/* the same in f64 (precision = "double": tcrossprod of a base matrix; the ridge still passes through fl()) */
int rsparse_hip_gramian_double(const double* X, int rank, int64_t n, double lambda, double* XtX_out);

/* replaces initialize_biases_float / initialize_biases_double (src/wrmf_init.cpp:5-34; .Call
 * _rsparse_initialize_biases_{float,double}, src/RcppExports.cpp:417-454): the 9 arguments of the reference with the
 * two S4 matrices flattened to their slots -- m_csc_r = users x items by item column (csc_p [n_items+1], csc_i, csc_x),
 * m_csr_r = the same matrix by user column (csr_p [n_users+1], csr_i, csr_x) -- plus the returned global bias.
 * user_bias (n_users) / item_bias (n_items) are read (initial values) and written.  With is_explicit_feedback and
 * calculate_global_bias the global mean is removed from csc_x and csr_x in place, as the reference does
 * (inst/include/wrmf_utils.hpp:41-52). */
int rsparse_hip_initialize_biases_float(int n_users, int n_items, const int32_t* csc_p, const int32_t* csc_i,
                                        double* csc_x, const int32_t* csr_p, const int32_t* csr_i, double* csr_x,
                                        float* user_bias, float* item_bias, double lambda, int dynamic_lambda,
                                        int non_negative, int calculate_global_bias, int is_explicit_feedback,
                                        double* global_bias_out);
int rsparse_hip_initialize_biases_double(int n_users, int n_items, const int32_t* csc_p, const int32_t* csc_i,
                                         double* csc_x, const int32_t* csr_p, const int32_t* csr_i, double* csr_x,
                                         double* user_bias, double* item_bias, double lambda, int dynamic_lambda,
                                         int non_negative, int calculate_global_bias, int is_explicit_feedback,
                                         double* global_bias_out);

/* ------------------------------------------------------------------------------------------------
 * (2) device-resident layer
 * ---------------------------------------------------------------------------------------------- */

/* A CSC matrix resident in HBM plus its launch schedule (row-length buckets).  Replaces the
 * non-owning MappedCSC view (inst/include/mapped_csc.hpp:8-29, src/utils.cpp:69-78). */
typedef struct rsparse_hip_csc rsparse_hip_csc;

/* upload from host dgCMatrix slots (values f64 -> f32 on the way) */
int rsparse_hip_csc_create_host(int n_rows, int n_cols, const int32_t* col_ptrs,
                                const int32_t* row_indices, const double* values,
                                rsparse_hip_csc** out);
/* adopt arrays that already live on the current device (not copied, not freed by destroy;
 * they must outlive the handle).  d_values are f32. */
int rsparse_hip_csc_create_device(int n_rows, int n_cols, const int32_t* d_col_ptrs,
                                  const int32_t* d_row_indices, const float* d_values,
                                  rsparse_hip_csc** out);
int rsparse_hip_csc_destroy(rsparse_hip_csc* m);

/* On-device ingest -- replaces the host-side second orientation of a fit,
 * c_iu = t_shallow(as.csr.matrix(c_ui)) (R/model_WRMF.R:184-191), and the per-column arma::conv_to of the
 * values (inst/include/wrmf_implicit.hpp:182-183): the caller hands over ONE orientation.
 *
 * rsparse_hip_csc_transpose_device: CSC (d_p int32[n_cols+1], d_i int32[nnz], d_x f32[nnz]) of an
 * n_rows x n_cols matrix -> CSC of its transpose (d_pt int32[n_rows+1], d_it int32[nnz], d_xt f32[nnz]), row
 * indices ascending inside every output column (stable counting sort by row index).  All pointers are device
 * pointers; outputs are caller-allocated.  A row index outside [0, n_rows) -> RSPARSE_HIP_ERR_INVALID.
 * rsparse_hip_values_to_float_device: d_dst[e] = (float)d_src[e] (dgCMatrix@x is f64 on the wire). */
int rsparse_hip_csc_transpose_device(int n_rows, int n_cols, const int32_t* d_p, const int32_t* d_i,
                                     const float* d_x, int32_t* d_pt, int32_t* d_it, float* d_xt, void* stream);
int rsparse_hip_values_to_float_device(int64_t n, const double* d_src, float* d_dst, void* stream);
/* frozen != 0: the caller promises that the values of the handle (its own copy, or the adopted device array) do not change until
 * the promise is withdrawn (frozen = 0) or the handle destroyed -- what holds for `c_ui@x` / `c_iu@x` during a fit with implicit
 * feedback (R/model_WRMF.R:184-191: the matrices are prepared once).  The half-iterations then take the statistics of the values
 * that the fp16 matrix-core kernels scale their operands by (max confidence, "some confidence < 1") from ONE scan per handle
 * instead of one per call (0.46 ms per half-iteration at 5e8 non-zeros).  Results do not depend on it.  Library calls that
 * change values through a handle (rsparse_hip_initialize_biases_explicit_device) require frozen = 0. */
int rsparse_hip_csc_freeze_values(rsparse_hip_csc* m, int frozen);

/* info_out: [0] n_rows, [1] n_cols, [2] nnz, [3] rows with more than [7] non-zeros ("long" rows),
 * [4] longest row, [5] non-zeros in long rows, [6] empty rows, [7] per-wave tile capacity (32),
 * [8..13] rows and [14..19] non-zeros per CG launch bucket, [20] launch-table id, [21] segments of the long rows that
 * are split across workgroups (0 = none is), [22..27] waves per row
 * (team size) of each bucket (0 = bucket unused), [28..33] resident quads (4 non-zeros) per wave,
 * [34..39] waves per workgroup (negative = the bucket streams rows longer than the resident capacity). */
int rsparse_hip_csc_info(const rsparse_hip_csc* m, int64_t info_out[40]);

/* XtX = X X^T + fl(lambda) I on the device (MFMA).  d_sumsq_out (nullable, device double[1])
 * receives sum(X^2) = trace before the ridge -- the `accu(X % X)` term of the loss
 * (inst/include/wrmf_implicit.hpp:299-301) for free.  stream: hipStream_t (NULL = default). */
int rsparse_hip_gramian_device(const float* d_X, int rank, int64_t n, double lambda,
                               float* d_XtX_out, double* d_sumsq_out, void* stream);

/* The same, and *d_absmax_inout = max(*d_absmax_inout, max |X|) (device float, nullable; the caller zeroes it before the
 * first block): the matrix is read here anyway.  What it is for: the long-row kernel of the implicit half-iteration
 * (wrmf_ne.hip) scales its fp16 operands by a power of two taken from max |X|, which it otherwise finds by scanning X
 * once per half-iteration call -- see d_absmax of rsparse_hip_als_implicit_device. */
int rsparse_hip_gramian_absmax_device(const float* d_X, int rank, int64_t n, double lambda, float* d_XtX_out,
                                      double* d_sumsq_out, float* d_absmax_inout, void* stream);

/* One implicit half-iteration over the columns of `conf` (als_implicit<float>, no-bias branch).
 * d_Y points at column 0 of this matrix's block (rank x n_cols).  Writes to d_loss_rows_out
 * (nullable, device double[1]) the un-normalised row part of the loss:
 *     sum_i [ sum_j c_ij (1 - y_i.x_j)^2 + lambda |y_i|^2 ]          (wrmf_implicit.hpp:259-261)
 * the caller adds lambda*sum(X^2) and divides by nnz (:286-304) -- kept separate so that shards
 * on several GPUs can be summed.  Asynchronous on `stream`.
 * d_absmax (nullable, device float[1]): max |X| if the caller knows it (rsparse_hip_gramian_absmax_device yields it; a
 * sharded driver passes the all-reduced maximum of its ranks' blocks and solves all its sub-blocks under it).  Any
 * value >= the true maximum is correct, a tight one is accurate; it must stay valid until the call has executed on
 * `stream`.  NULL: the library scans X (0.5 ms per GB).  Per call, no state is kept. */
int rsparse_hip_als_implicit_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y,
                                    const float* d_XtX, int rank, double lambda, unsigned solver,
                                    unsigned cg_steps, const float* d_absmax, double* d_loss_rows_out, void* stream);

/* One explicit half-iteration (als_explicit<float>, no-bias branch).  Loss row part:
 *     sum_i [ sum_j (r_ij - y_i.x_j)^2 + lambda_use_i |y_i|^2 ]      (wrmf_explicit.hpp:131-132) */
int rsparse_hip_als_explicit_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y,
                                    int rank, double lambda, unsigned solver, unsigned cg_steps,
                                    int dynamic_lambda, double* d_loss_rows_out, void* stream);

/* als_implicit<T> with_biases = TRUE, global_bias = 0, Cholesky or NNLS (inst/include/wrmf_implicit.hpp:114-154,
 * 186-252,256-270), device-resident form.  Layout of X / Y as for the explicit variant below.  d_XtX is the
 * (rank-1) x (rank-1) Gramian of X without its x_bias row, ridge included (R/model_WRMF.R:463-486; use
 * rsparse_hip_gramian_device on the re-packed matrix).  Every row is solved, empty ones too (:178).  solver =
 * conjugate_gradient -> RSPARSE_HIP_ERR_UNSUPPORTED: the reference drops a row of the warm start twice on that path
 * (:189,197) and cannot run it.  The regulariser on X (all rows but the ones, :287-297) is the caller's. */
int rsparse_hip_als_implicit_bias_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y,
                                         const float* d_XtX, int rank, double lambda, unsigned solver,
                                         int is_x_bias_last_row, double* d_loss_rows_out, void* stream);

/* als_implicit<T> with a global bias (inst/include/wrmf_implicit.hpp:108-112,146-157,228-229,262-270),
 * device-resident form; with_biases selects the user/item-bias layout of rsparse_hip_als_implicit_bias_device (then
 * rhs_init = -X' (x_b + global_bias), :152; Cholesky / NNLS only), otherwise X / Y / XtX are the plain rank x n matrices
 * and every right-hand side gets global_bias_base = -global_bias * rowSums(X) (:111-112, computed on the device); every
 * column is solved, empty ones too (:178).  The loss compares x_j.y with 1 - global_bias (- x_b).  solver =
 * conjugate_gradient (no biases): cg_solver_implicit_global_bias (:35-57, 203) from the warm start in d_Y, cg_steps
 * steps.  global_bias itself is the caller's: sum(x) / (sum(x) + n_user n_item - nnz), R/model_WRMF.R:286-287.
 * double_threshold: which build's cut-off applies to a small global bias (:108-109) -- 0 = als_implicit<float>'s
 * sqrt(FLT_EPSILON) = 3.45e-4, non-zero = als_implicit<double>'s 1.49e-8 (a model declared with precision = "double" whose
 * arithmetic runs on this fp32 layer: on large sparse data sum / (sum + n_user n_item - nnz) is typically below 3.45e-4,
 * and the double build keeps it).  d_absmax: as for rsparse_hip_als_implicit_device. */
int rsparse_hip_als_implicit_global_bias_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y,
                                                const float* d_XtX, int rank, double lambda, unsigned solver,
                                                unsigned cg_steps, int with_biases, int is_x_bias_last_row,
                                                double global_bias, int double_threshold, const float* d_absmax,
                                                double* d_loss_rows_out, void* stream);

/* initialize_biases_implicit (inst/include/wrmf_utils.hpp:86-165; .Call _rsparse_initialize_biases_{double,float} with
 * is_explicit_feedback = FALSE).  calculate_global_bias: sum(x) / (sum(x) + n_users n_items - nnz) (:90-93), subtracted
 * inside the sweeps (:142,157) and returned through global_bias_out (may be NULL). */
int rsparse_hip_initialize_biases_implicit_device(const rsparse_hip_csc* c_ui, const rsparse_hip_csc* c_iu,
                                                  float* d_user_bias, float* d_item_bias, double lambda,
                                                  int non_negative, int calculate_global_bias,
                                                  double* global_bias_out, void* stream);

/* als_explicit<T> with_biases = TRUE (inst/include/wrmf_explicit.hpp:41-64,86-91,113-127), device-resident form.
 * rank counts the two extra coordinates (R/model_WRMF.R:160: rank + 2): X = [1, ..., x_bias] and
 * Y = [y_bias, ..., 1] when is_x_bias_last_row, X = [x_bias, ..., 1] and Y = [1, ..., y_bias] otherwise; the
 * placeholder entry of every Y row is left untouched.  d_loss_rows_out as for rsparse_hip_als_explicit_device;
 * the regulariser on X skips the row of ones (:147-159) and is the caller's (rsparse_hip_weighted_sumsq_device on
 * the other rank-1 rows). */
int rsparse_hip_als_explicit_bias_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y, int rank,
                                         double lambda, unsigned solver, unsigned cg_steps, int dynamic_lambda,
                                         int is_x_bias_last_row, double* d_loss_rows_out, void* stream);

/* with_global_bias without user/item biases, explicit feedback (R/model_WRMF.R:278-282): global_bias = mean(c_ui@x),
 * removed in place from the resident values of both orientations (d_x_other may be NULL).  *mean_out is a host double. */
int rsparse_hip_values_subtract_mean_device(int64_t n, float* d_x, float* d_x_other, double* mean_out, void* stream);

/* initialize_biases_explicit (inst/include/wrmf_utils.hpp:32-84; .Call _rsparse_initialize_biases_{double,float} with
 * is_explicit_feedback = TRUE, src/RcppExports.cpp:417-454).  c_ui: users x items by item column, c_iu: its transpose.
 * With calculate_global_bias the mean of the values is removed from the resident values of BOTH handles in place (as
 * the reference does to ConfCSC / ConfCSR) and returned in *global_bias_out (host).  d_user_bias [n_users] (read as
 * the starting point, the R driver passes zeros) and d_item_bias [n_items] are device vectors. */
int rsparse_hip_initialize_biases_explicit_device(rsparse_hip_csc* c_ui, rsparse_hip_csc* c_iu, float* d_user_bias,
                                                  float* d_item_bias, double lambda, int dynamic_lambda,
                                                  int non_negative, int calculate_global_bias,
                                                  double* global_bias_out, void* stream);

/* The bias initialisation one sweep at a time, over ONE block of columns: for drivers that shard the matrix (every rank
 * owns a block of items and a block of users and holds both bias vectors in full, exchanging the swept block after each
 * sweep).  initialize_biases_explicit (inst/include/wrmf_utils.hpp:54-82) is five times
 *     item sweep: item_bias[c] = sum_{e in c} (x_e - user_bias[idx_e]) / (lambda_use + n_c)     then the same for the users;
 * initialize_biases_implicit (:86-165) is, once, means / adjustments per column (prep; n_other = the TRUE size of the other
 * side), then five times an item sweep and a user sweep of the weighted running mean (:136-143, 152-159), each given the
 * SUM of the other side's current biases (d_other_sum, device double[1]; NULL = 0: the first item sweep, :131-135).
 * d_other_bias is indexed by conf's row indices; d_out / d_means / d_adj by conf's columns. */
int rsparse_hip_bias_sweep_explicit_device(const rsparse_hip_csc* conf, const float* d_other_bias, double lambda,
                                           int dynamic_lambda, int non_negative, float* d_out, void* stream);
int rsparse_hip_bias_prep_implicit_device(const rsparse_hip_csc* conf, int n_other, double lambda, double* d_means,
                                          double* d_adj, void* stream);
int rsparse_hip_bias_sweep_implicit_device(const rsparse_hip_csc* conf, const float* d_other_bias, int n_other,
                                           const double* d_other_sum, const double* d_means, const double* d_adj,
                                           int non_negative, double global_bias, float* d_out, void* stream);

/* sum_j w_j |X[:,j]|^2 on the device (w = NULL -> 1): the regulariser terms
 * lambda*accu(X%X) and lambda*accu((X%X)*cnt_X) (wrmf_explicit.hpp:160-170). */
int rsparse_hip_weighted_sumsq_device(const float* d_X, int rank, int64_t n, const float* d_w,
                                      double* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * `$predict`: top-k of the dense product (next step after the solve on the same operator boundary)
 * ---------------------------------------------------------------------------------------------- */

#define RSPARSE_HIP_NA_INTEGER INT32_MIN /* R's NA_integer_: fewer than k admissible items */
#define RSPARSE_HIP_MAX_TOPK 256

/* replaces top_product (src/matrix_top_product.cpp:20-102; .Call `_rsparse_top_product`,
 * R/RcppExports.R).  x: nr x rank and y: rank x nc, both column-major doubles as arma::mat holds them;
 * not_recommend as dgRMatrix slots p (nr+1) / j (sorted per row), NULL = nothing to filter; exclude: 1-based
 * item indices excluded for every row.  res: nr x k column-major 1-based indices (NA_integer_ where fewer
 * than k items are admissible), scores: nr x k column-major (+ glob_mean), best first; equal scores keep the
 * reference's order (larger index first).  The candidates (k + max(8, k / 4) per row) come from an fp32 matrix-core pass,
 * their scores are recomputed in double from x and y as given and the reference's heap is replayed over them
 * (rsparse_hip_top_product_f64_device below).  k > 256 -> ERR_UNSUPPORTED.  n_threads is accepted and ignored. */
int rsparse_hip_top_product(const double* x, const double* y, int nr, int nc, int rank, unsigned k,
                            unsigned n_threads, const int32_t* not_recommend_p,
                            const int32_t* not_recommend_j, const int32_t* exclude, int n_exclude,
                            double glob_mean, int32_t* res, double* scores);

/* device-resident form: d_U n_users x rank and d_V n_items x rank row-major fp32 (= rank x n column-major),
 * d_exclude0: sorted 0-based item indices, d_res / d_scores: n_users x k row-major.
 * A call for more than 128 users (rank <= 128) keeps its per-user candidate buffers in the library's grow-only workspace:
 * min(n_users, 131072) x 2 x (k + 32 + max(64, k)) words, 121 MB at k = 10, 570 MB at k = 256; longer calls run in chunks of
 * 131072 users on the stream and reuse it. */
int rsparse_hip_top_product_device(const float* d_U, const float* d_V, int n_users, int n_items, int rank,
                                   int k, const int32_t* d_not_recommend_p, const int32_t* d_not_recommend_j,
                                   const int32_t* d_exclude0, int n_exclude, double glob_mean,
                                   int32_t* d_res, float* d_scores, void* stream);

/* `$predict` that ORDERS like the reference.  find_top_product casts both factor matrices to double before the product
 * (R/utils.R:35-36) and top_product takes arma::mat (src/matrix_top_product.cpp:20): the fp32 pass above only nominates -- it keeps
 * the k + extra best items of every user (extra < 0: max(8, k / 4); never more than 256 candidates) --, their scores are
 * recomputed in double from d_U64 / d_V64 (n x rank row-major doubles; both NULL: from the fp32 factors widened) and the
 * reference's heap is replayed over the candidates in ascending item order: strict `>` replacement, equal scores with the
 * larger index first, and when more candidates sit AT the k-th score than places, the ones the reference's heap keeps.
 * Exact whenever every item whose double score reaches the k-th best is among the candidates.  d_scores: n_users x k doubles. */
int rsparse_hip_top_product_f64_device(const float* d_U, const float* d_V, const double* d_U64, const double* d_V64,
                                       int n_users, int n_items, int rank, int k, int extra,
                                       const int32_t* d_not_recommend_p, const int32_t* d_not_recommend_j,
                                       const int32_t* d_exclude0, int n_exclude, double glob_mean, int32_t* d_res,
                                       double* d_scores, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (3) fp64 device layer: als_implicit<double> / als_explicit<double> with the data resident in HBM
 * ---------------------------------------------------------------------------------------------- */

/* What the `*_double` .Call targets compute (src/wrmf_implicit.cpp:5-14, src/wrmf_explicit.cpp:5-14), device-resident
 * like layer (2): CSC with f64 values, X / Y / XtX column-major f64.  One kernel family covers every variant -- implicit
 * and explicit feedback, Cholesky (general-solver fallback included) / conjugate gradient / NNLS, user/item biases, the
 * implicit global bias -- by assembling each row's system in LDS (wrmf_f64.hip); it is the parity path of
 * precision = "double", not the bench path. */
typedef struct rsparse_hip_csc_f64 rsparse_hip_csc_f64;

/* adopt arrays that live on the current device (not copied, not freed; they must outlive the handle); validated like
 * rsparse_hip_csc_create_device */
int rsparse_hip_csc_f64_create_device(int n_rows, int n_cols, const int32_t* d_col_ptrs, const int32_t* d_row_indices,
                                      const double* d_values, rsparse_hip_csc_f64** out);
int rsparse_hip_csc_f64_destroy(rsparse_hip_csc_f64* m);
/* Conjugate gradient in double: a row of more than min_len non-zeros is cut into chunks of chunk_len that run as waves of their
 * own, pass by pass (wrmf_f64.hip, "long rows": one wave per row left a half-iteration waiting for its longest row; sums in
 * chunk order, no atomics).  Defaults 2048 / 1024 (0 = restore the default); applies to the handles made -- and the stateless
 * *_double calls issued -- afterwards.  Results do not depend on it beyond rounding (1e-12); the tests lower it so that small
 * matrices take the path. */
int rsparse_hip_set_f64_long_rows(int min_len, int chunk_len);

/* XtX = X X^T + fl(lambda) I in f64 (the ridge is rounded to fp32 in the double build too: float::fl(diag(lambda)),
 * R/model_WRMF.R:476); d_sumsq_out (nullable) = sum(X^2) */
int rsparse_hip_gramian_f64_device(const double* d_X, int rank, int64_t n, double lambda, double* d_XtX_out,
                                   double* d_sumsq_out, void* stream);

/* One half-iteration in f64: als_implicit<double> (implicit != 0; inst/include/wrmf_implicit.hpp:90-305) or
 * als_explicit<double> (wrmf_explicit.hpp:33-174) over the columns of `conf`.  Arguments as in layer (2):
 * with_biases / is_x_bias_last_row select the user/item-bias layout (rank counts the two extra coordinates; d_XtX is
 * then (rank-1) x (rank-1)); global_bias (implicit feedback; below sqrt(DBL_EPSILON) = none, :108-109) is handled with
 * every solver, global_bias_base = -global_bias * rowSums(X) computed on the device; dynamic_lambda: explicit feedback
 * only.  implicit + with_biases + conjugate_gradient -> RSPARSE_HIP_ERR_UNSUPPORTED (:189,197).  d_loss_rows_out
 * (nullable, device double[1]): the row part of the loss as for rsparse_hip_als_{implicit,explicit}_device.  A system
 * that is not positive definite is re-solved by Gaussian elimination with partial pivoting inside the kernel and counted
 * in rsparse_hip_take_numeric_failures. */
int rsparse_hip_als_f64_device(const rsparse_hip_csc_f64* conf, int implicit, const double* d_X, double* d_Y,
                               const double* d_XtX, int rank, double lambda, unsigned solver, unsigned cg_steps,
                               int dynamic_lambda, int with_biases, int is_x_bias_last_row, double global_bias,
                               double* d_loss_rows_out, void* stream);

/* initialize_biases_double (src/wrmf_init.cpp:5-19 -> inst/include/wrmf_utils.hpp:32-165), device-resident: c_ui =
 * users x items by item column, c_iu = its transpose.  With is_explicit_feedback and calculate_global_bias the mean of
 * the values is removed from the resident values of BOTH handles in place. */
int rsparse_hip_initialize_biases_f64_device(rsparse_hip_csc_f64* c_ui, rsparse_hip_csc_f64* c_iu, double* d_user_bias,
                                             double* d_item_bias, double lambda, int dynamic_lambda, int non_negative,
                                             int calculate_global_bias, int is_explicit_feedback,
                                             double* global_bias_out, void* stream);

/* f64 counterparts of the single bias sweeps (rsparse_hip_bias_*_device above) */
int rsparse_hip_bias_sweep_explicit_f64_device(const rsparse_hip_csc_f64* conf, const double* d_other_bias, double lambda,
                                               int dynamic_lambda, int non_negative, double* d_out, void* stream);
int rsparse_hip_bias_prep_implicit_f64_device(const rsparse_hip_csc_f64* conf, int n_other, double lambda, double* d_means,
                                              double* d_adj, void* stream);
int rsparse_hip_bias_sweep_implicit_f64_device(const rsparse_hip_csc_f64* conf, const double* d_other_bias, int n_other,
                                               const double* d_other_sum, const double* d_means, const double* d_adj,
                                               int non_negative, double global_bias, double* d_out, void* stream);

/* f64 counterparts of rsparse_hip_values_subtract_mean_device / rsparse_hip_weighted_sumsq_device */
int rsparse_hip_values_subtract_mean_f64_device(int64_t n, double* d_x, double* d_x_other, double* mean_out, void* stream);
int rsparse_hip_weighted_sumsq_f64_device(const double* d_X, int rank, int64_t n, const double* d_w, double* d_out,
                                          void* stream);

/* Kernel timing for measurement harnesses (bench.py): when enabled, every device-layer call brackets
 * its kernels with HIP events on the caller's stream.  rsparse_hip_profile_last() waits for the last
 * call and returns milliseconds per segment in launch order:
 *   CG half-iterations -> [0..5] the launches of the row-length buckets of rsparse_hip_csc_info ([0] = the
 *     normal-equation kernel for the rows beyond 512 non-zeros), [6] loss reduction (with the LDS-tile fallback kernels
 *     for ranks that are not a multiple of 4: [0] short-row, [1] long-row, [2] loss);
 *   Cholesky -> [0] the normal-equation launch with the exact solve (long rows), [1] the low-rank kernel (short rows,
 *     incl. its one-workgroup preparation), [2] the k x k kernel, [3] loss reduction;
 *   NNLS -> [0] kernel, [2] loss reduction;   Gramian -> [0] MFMA partial kernel, [1] reduction.
 * rsparse_hip_profile_last_names() gives, for the same call, the name of the kernel each segment timed -- newline
 * separated, an empty line for a segment that launched nothing -- taken from the runtime's own symbol table
 * (hipKernelNameRefByPtr, demangled), i.e. exactly what rocprofv3 prints for it. */
int rsparse_hip_profile_enable(int on);
int rsparse_hip_profile_last_names(char* buf, int cap);
/* How the launches of one conjugate-gradient half-iteration (one per row-length bucket, disjoint rows) are issued:
 * 2 (default) = the long-row launch on the caller's stream, the others on side streams forked from / joined to it,
 * 1 = every launch on a side stream, 0 = all back to back on the caller's stream -- what a profiler needs for
 * well-defined per-kernel durations (bench.py --serial-launches).  Results do not depend on it. */
int rsparse_hip_set_launch_mode(int mode);
int rsparse_hip_profile_last(double ms_out[8]);

/* The exact (Cholesky) solver's bookkeeping since the last call of this function; resets.  A per-row system whose
 * factorisation meets a non-positive pivot is re-solved on the device by Gaussian elimination with partial pivoting -- what
 * arma::solve(lhs, rhs, fast + likely_sympd) falls back to behind a warning (inst/include/wrmf_implicit.hpp:236,
 * wrmf_explicit.hpp:108): *fallback_out (nullable) = the number of such rows, the reference's warnings.  *unresolved_out =
 * the rows whose general solve failed too (an exactly singular system; their solution was set to zero): the half-iteration
 * entry points report them as RSPARSE_HIP_ERR_NUMERIC (the stateless ones at once, the device-resident ones through this
 * call).  Reads device memory: synchronises. */
int rsparse_hip_take_numeric_failures(int64_t* unresolved_out, int64_t* fallback_out);

/* ------------------------------------------------------------------------------------------------
 * (4) multi-GPU context: the sharded driver inside the library (ABI version 6)
 * ---------------------------------------------------------------------------------------------- */

/* What the reference's host calls per half-iteration is ONE function that internally drives its OpenMP threads
 * (`als_implicit_double(c_ui, X, Y, XtX, ...)`, R/model_WRMF.R:111-147 -> src/RcppExports.cpp:371-415,
 * inst/include/wrmf_implicit.hpp:162-175 `#pragma omp parallel for`); this layer is the same contract over 1..8 GPUs: the caller
 * stays ONE host thread (an R session), the library runs one persistent host thread per device, shards users and items over
 * them in contiguous nnz-balanced blocks, keeps full factor replicas, and per half-iteration issues one fused exchange of the
 * k x k Gramian partials (+ sum(F^2), max |F|), an in-place all-gather of every solved sub-block (overlapped with the next
 * sub-block's solve on a second stream) and one all-reduce of the loss terms -- SURVEY.md 8(e).  The per-rank arithmetic is
 * layer (2)'s, row for row: results do not depend on the number of ranks except through the summation order of the Gramian
 * and of the loss (as the reference's do not depend on OMP_NUM_THREADS except through its loss reduction).
 *
 * comm_kind: RSPARSE_HIP_COMM_RCCL -- one device per rank, collectives = RCCL over xGMI (ncclCommInitAll; librccl is loaded
 * with dlopen when the context is created, a single-GPU user never needs it); RSPARSE_HIP_COMM_SHARED -- the ranks are threads
 * with streams of their own on ONE device (device_ids may repeat; default: all on device 0) and a collective is a host
 * barrier + device copies: the transport of the tests and of single-GPU dry runs, everything else is the production code.
 * device_ids: n_ranks entries, or NULL (RCCL: rank r on device r). */
typedef struct rsparse_hip_ctx rsparse_hip_ctx;
#define RSPARSE_HIP_COMM_RCCL 0
#define RSPARSE_HIP_COMM_SHARED 1
#define RSPARSE_HIP_SIDE_ITEMS 0 /* solve the item factors given the user factors (R/model_WRMF.R:321) */
#define RSPARSE_HIP_SIDE_USERS 1 /* solve the user factors given the item factors (:327) */
int rsparse_hip_ctx_create(int n_ranks, const int* device_ids, int comm_kind, rsparse_hip_ctx** out);
int rsparse_hip_ctx_destroy(rsparse_hip_ctx* ctx);

/* The interaction matrix in both orientations as the R driver holds them (R/model_WRMF.R:184-191): c_ui = users x items as
 * dgCMatrix slots (ui_p int32[n_item + 1], ui_i = user ids, ui_x), c_iu = its transpose (iu_p int32[n_user + 1], iu_i = item
 * ids, iu_x); host arrays, 0-based, row indices ascending inside a column.  Every rank uploads its own blocks only.
 * n_sub_users / n_sub_items: sub-blocks per rank and half-iteration (0 = default: 1 for one rank, else 4). */
int rsparse_hip_ctx_set_matrix(rsparse_hip_ctx* ctx, int n_user, int n_item, const int32_t* ui_p, const int32_t* ui_i,
                               const double* ui_x, const int32_t* iu_p, const int32_t* iu_i, const double* iu_x,
                               int n_sub_users, int n_sub_items);

/* Factor matrices as the reference passes them: U = rank x n_user, V = rank x n_item, column-major fp32 (every entity's vector
 * contiguous).  set: host -> every replica; get (either pointer may be NULL): rank 0's replica -> host. */
int rsparse_hip_ctx_set_factors(rsparse_hip_ctx* ctx, int rank, const float* U, const float* V);
int rsparse_hip_ctx_get_factors(rsparse_hip_ctx* ctx, float* U, float* V);

/* One half-iteration over all devices: als_implicit<float> / als_explicit<float>, no-bias branch, every solver of layer (2)
 * (the exact solve that ends a fit -- R/model_WRMF.R:355-359 -- is side = USERS with solver = cholesky after set_factors of
 * zeros for U).  implicit: the Gramian XtX + fl(lambda) I of the fixed side is formed inside (R/model_WRMF.R:474-486).
 * *loss_out (nullable) = the loss as the reference reports it: (row terms + lambda * regulariser) / nnz
 * (wrmf_implicit.hpp:286-304, wrmf_explicit.hpp:146-173).  Synchronous. */
int rsparse_hip_ctx_half_iteration(rsparse_hip_ctx* ctx, int side, int implicit, double lambda, unsigned solver,
                                   unsigned cg_steps, int dynamic_lambda, double* loss_out);

/* rsparse_hip_take_numeric_failures summed over the ranks (one decision for the whole context). */
int rsparse_hip_ctx_take_numeric_failures(rsparse_hip_ctx* ctx, int64_t* unresolved_out, int64_t* fallback_out);

/* info_out: [0] ranks, [1] comm_kind, [2] n_user, [3] n_item, [4] nnz, [5] rank of the factors, [6] / [7] sub-blocks per rank
 * (users / items), [8] / [9] rows per sub-block, [10] / [11] users / items owned by rank 0, [12] 1 = the collectives are RCCL.
 * times_out (nullable): of the last half-iteration, the slowest rank's [0] wall milliseconds, [1] milliseconds inside
 * collective calls (host side: issue time for RCCL, the whole exchange for SHARED). */
int rsparse_hip_ctx_info(const rsparse_hip_ctx* ctx, int64_t info_out[16], double times_out[2]);

#ifdef __cplusplus
}
#endif
#endif /* RSPARSE_WRMF_HIP_H */
