// Internal interface between the C-ABI layer (wrmf_capi.cpp) and the kernel launchers
// (wrmf_kernels.hip).  Not installed; the public boundary is include/rsparse_wrmf_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

namespace rsparse_hip {

// Device-resident CSC + launch schedule.
struct DevCSC {
  int n_rows = 0;  // = number of entities on the fixed side (columns of X)
  int n_cols = 0;  // = rows solved
  int64_t nnz = 0;
  const int32_t* col_ptrs = nullptr;
  const int32_t* row_idx = nullptr;
  const float* vals = nullptr;
  // rows with more than `short_max` non-zeros, longest first: solved one workgroup per row
  int32_t* long_rows = nullptr;
  int n_long = 0;
  int short_max = 0;  // tile capacity T the schedule was built for
  int max_len = 0;
  // quad-layout CG schedule: every row, longest first; bucket b occupies order[q_off[b], q_off[b+1])
  // (bucket table: wrmf_cgq.hip kBuckets)
  int32_t* q_order = nullptr;
  int q_off[7] = {0, 0, 0, 0, 0, 0, 0};
  int q_cfg = 0;
  int q_pair_first = 0;   // position in q_order of the first row with at most 16 non-zeros (wrmf_cgp.hip: two rows per wave)
  int64_t q_nnz[6] = {0, 0, 0, 0, 0, 0};
  int64_t* q_stream_off = nullptr;  // prefix sums of the streamed bucket's row lengths (device)
  // normal-equation kernel (wrmf_ne.hip): the long rows (bucket 0) dealt to q_ne_wg workgroups, longest processing
  // time first; workgroup b owns q_ne_rows[q_ne_ptr[b], q_ne_ptr[b+1])
  int q_n_chol_long = 0;   // rows of more than kCholLongLen non-zeros (a prefix of q_order)
  int q_gt32 = 0;                   // rows of more than 32 non-zeros
  int q_gt48 = 0;                   // ... of more than 48
  int q_lr_first = 0, q_n_lr = 0;   // rows of 1..kCholLrMax non-zeros: q_order[q_lr_first, q_lr_first + q_n_lr)
  int32_t* q_ne_rows = nullptr;
  int32_t* q_ne_ptr = nullptr;
  int q_ne_wg = 0;
  int32_t* q_ne1_rows = nullptr;   // the same rows dealt to one list per workgroup slot (kernels resident once per CU)
  int32_t* q_ne1_ptr = nullptr;
  int q_ne1_wg = 0;
  int32_t* q_ne_segs = nullptr;   // [q_ne_nseg][6]: segments of the rows split across workgroups (wrmf_capi.cpp)
  int q_ne_nseg = 0;
  int q_ne_entries = 0;           // list entries = rows that are not split + segments
  int32_t* q_ne_split_rows = nullptr;   // lists of the COLLECT launch (one workgroup per split row)
  int32_t* q_ne_split_ptr = nullptr;
  int q_ne_nsplit = 0;
  // the same lists for solver == CHOLESKY, whose normal-equation launch takes the rows beyond q_nec_min non-zeros (a
  // longer prefix of q_order, dealt with the larger fixed cost of the exact solve); they alias the lists above when the
  // two thresholds coincide
  int q_nec_min = 0;
  int q_n_nec = 0;      // rows of more than q_nec_min non-zeros (a prefix of q_order)
  bool q_nec_own = false;
  int32_t* q_nec_rows = nullptr;
  int32_t* q_nec_ptr = nullptr;
  int q_nec_wg = 0;
  int32_t* q_nec_segs = nullptr;
  int q_nec_nseg = 0;
  int q_nec_entries = 0;
  int32_t* q_nec_split_rows = nullptr;
  int32_t* q_nec_split_ptr = nullptr;
  int q_nec_nsplit = 0;
  int64_t nnz_long = 0;
  int n_empty = 0;
  bool owns_matrix = false;
  // rsparse_hip_csc_freeze_values: the caller promises that the values do not change while the flag is set; the statistics of
  // the values that the fp16 kernels scale their operands by (max c, "some c < 1": launch_ne_stats) are then scanned once per
  // handle instead of once per half-iteration.  vstats: 2 words on the device, valid once vstats_valid
  mutable bool vals_frozen = false;
  mutable bool vstats_valid = false;
  mutable unsigned* vstats = nullptr;
  mutable hipStream_t vstats_stream = nullptr;   // the stream whose scan wrote vstats: only work queued on it is ordered behind that scan
};

struct AlsArgs {
  const int32_t* col_ptrs;
  const int32_t* row_idx;
  const float* vals;
  const float* X;    // k x n_rows, ld = k
  float* Y;          // k x n_cols, ld = k
  const float* XtX;  // k x k (implicit) or nullptr
  const int32_t* long_rows;
  int n_long;
  int n_cols;
  int k;
  int cg_steps;
  float lambda;        // lambda as the kernels use it in fp32 (regulariser inside the explicit system)
  double lambda_loss;  // lambda of the loss term (double, wrmf_implicit.hpp:259-261)
  int dynamic_lambda;
  double* loss_partials;  // one double per wave (short kernel) / per workgroup (long kernel)
  // Cholesky: fail_counter[0] = rows whose factorisation met a non-positive pivot (they are appended to fail_rows, at most
  // fail_cap of them, and re-solved by the general solver, wrmf_lu.hip), fail_counter[1] = rows that one could not solve either
  int* fail_counter;
  int* fail_rows;
  int fail_cap;
  // streamed CG rows: scratch [cg_steps+1][stream_nnz] for the per-non-zero dot products of every sweep,
  // stream_off[r] = first slot of the r-th streamed row (rows in schedule order); nullptr = re-gather for the loss
  const float* zero_row;  // >= 128 zero floats: what the padding slots of a gather read
  // user/item biases with implicit feedback (Cholesky / NNLS kernels only; all nullptr otherwise):
  const float* rhs_vals;   // per non-zero: coefficient of x_j in the right-hand side, c - x_b (c - 1)  (default: vals)
  const float* loss_tgt;   // per non-zero: target of the loss term, 1 - global_bias - x_b               (default: loss_tgt_const)
  float loss_tgt_const;    // target of the loss term when loss_tgt is nullptr: 1, or 1 - global_bias (wrmf_implicit.hpp:262-264)
  const float* rhs_init;   // k floats added to every right-hand side; non-null also means "solve empty rows too"
  // implicit feedback, conjugate gradient with a global bias (cg_solver_implicit_global_bias, wrmf_implicit.hpp:35-57):
  // gbias != 0 selects it; rhs_init = global_bias_base, loss_tgt_const = 1 - gbias.  The normal-equation kernel's rows
  // (wrmf_ne.hip) take the extra term of their first residual, global_bias_base - gbias X_nnz (c - 1), from ne_r0
  // [ne_r0_slot[row]][k] (launch_gb_row_terms)
  float gbias;
  const float* ne_r0;
  const int32_t* ne_r0_slot;
  float* tscr;
  const int64_t* stream_off;
  int64_t stream_nnz;
  // Cholesky: rows of more than kCholLongLen non-zeros = the first n_chol_long entries of the length-sorted row order
  const int32_t* chol_long_rows;
  int n_chol_long;
  // ... and the main launch's rows as ranges of the same order (nullptr: it walks every column and skips what it does
  // not own): [chol_first, chol_first + chol_n_main) = the rows of more than kCholLrMax non-zeros that no other launch
  // takes, then, when the low-rank kernel stands down, the short rows behind them; the empty rows
  // [chol_empty_first, n_cols) always
  const int32_t* chol_list;
  int chol_first, chol_n_main, chol_empty_first;
  // Cholesky, short rows: lr_rows = the n_lr rows of 1..kCholLrMax non-zeros; lr_flags (device word, nullable): 0 = the
  // low-rank kernel solves them and wrmf_chol.hip's skips them, else the other way round; lr_M = 3 x 128 x 128 floats
  const int32_t* lr_rows;
  int n_lr;
  int lr_n_gt32, lr_n_gt16;   // how many of them (a prefix: the list is longest first) have more than 32 / 16 non-zeros; -1 = unknown
  int lr_n_gt48;              // ... more than 48
  float* nnls_lhs;            // NNLS, rank 65..128: chol_loss_slots(n_cols) x 128 x 128 floats -- lhs of the workgroup's row (nullable: lhs in LDS)
  const int32_t* nnls_order;  // NNLS, one wave per row: the rows longest first (nullable: natural order) -- a row's cost is its sweep count,
                              // which grows with its length, and what runs last decides when the launch ends
  int lrx;                    // explicit feedback: lr_rows are solved by the push-through wave kernel (wrmf_chol_lr.hip) and the k x k kernels skip them
  unsigned* lr_flags;
  float* lr_M;
  // rows split across workgroups: segment table, per-segment partial accumulators (kNeSegFloats floats each) and flags
  const int32_t* ne_segs;
  float* ne_seg_scratch;
  int* ne_seg_flags;
  int ne_chol;                   // long rows through wrmf_ne.hip with the exact solve instead of CG (solver == CHOLESKY)
  int ne_chol_min;               // ... the rows of more than this many non-zeros (wrmf_chol.hip skips them)
  const unsigned* ne_stats;      // implicit NE launches: {bits of max |x|, bits of max c, any c < 1} (launch_ne_stats), or nullptr
  const float* mf_XtX;           // wrmf_chol_mf.hip, implicit feedback: XtX padded to 128 x 128 (identity beyond the rank); = XtX at rank 128
  const unsigned* wave_stats;    // the same block for the wave-per-row kernels at rank 33..64 (operand scales of their matrix-core assembly), or nullptr
  unsigned long long* ne_prof;   // RSP_NE_PROF builds: [workgroup][wave][8] cycle counters of wrmf_ne.hip (else nullptr)
};

struct QSchedule {
  const int32_t* order;
  int off[7];
  int pair_first;   // see DevCSC::q_pair_first
  int cfg;  // geometry the schedule was built for (see wrmf_cgq.hip kBuckets)
  const int32_t* ne_rows;  // see DevCSC::q_ne_*
  const int32_t* ne_ptr;
  int ne_wg;
  int ne_entries;                 // list entries = rows that are not split + segments
  const int32_t* ne_split_rows;   // per split row: -(index of its first segment + 1); ne_split_ptr = 0, 1, 2, ...
  const int32_t* ne_split_ptr;
  int ne_nsplit;
  const int32_t* mf_rows = nullptr;   // wrmf_cg_mf.hip's rows of the first bucket (the normal-equation lists above then hold the giant rows only)
  int mf_n = 0;
};
int cgq_default_cfg();
void cgq_set_launch_mode(int mode);   // rsparse_hip_set_launch_mode
int cgq_num_buckets();               // 6
int cgq_bucket_wpr(int cfg, int b);  // waves per row of bucket b (0 = unused)
int cgq_bucket_capq(int cfg, int b);
int cgq_bucket_waves(int cfg, int b);  // waves per workgroup of bucket b's kernel
int cgq_bucket_stream(int cfg, int b);
int cgq_bucket_grid(int n_rows, int bucket, int cfg);
int cgq_bucket_of(int len, int cfg);
size_t cgq_loss_slots(const QSchedule& q, int k, bool implicit);
// the rows of at most 16 non-zeros of the last bucket, two per wave (wrmf_cgp.hip): rank 65..128, implicit feedback
bool cgp_supported(int k, bool implicit);
int cgp_grid(int n_rows);
hipError_t launch_als_cgp(const AlsArgs& a, const int32_t* rows, int n_rows, size_t loss_slot0, hipStream_t s,
                          hipEvent_t* ev_slot);
// long rows (bucket 0) by one-pass normal equations on the matrix cores (wrmf_ne.hip) instead of the streamed CG kernel
bool ne_supported(int k);
constexpr int kNeMinLen = 512;       // its rows: more non-zeros than the largest resident bucket of wrmf_cgq.hip holds
constexpr int kNeCholMinLen = 64;   // (rounds 2-5: solver == CHOLESKY's threshold of the same launch; dev builds: RSPARSE_HIP_NE_CHOL_MIN)
constexpr int kNeMaxSeg = 16;        // segments per split row
constexpr int kNeMaxSegTotal = 1024;   // ... per list set (186 MB of partial accumulators at most)
constexpr int kNeSegFloats = 4 * (11 * 16 * 64 + 128 + 2);   // per segment: 4 waves x (<= 11 accumulator tiles + b + sum c)
// absmax_hint (nullable, device float): max |X| supplied by the caller -- X is then not scanned
// cached_vstats (nullable, 2 device words = stats[1..2] of an earlier scan of the same values): the values are not read;
// save_vstats (nullable): receives stats[1..2] of this scan
hipError_t launch_ne_stats(const float* X, int64_t nx, const float* vals, int64_t nnz, unsigned* stats, hipStream_t s,
                           const float* absmax_hint = nullptr, const unsigned* cached_vstats = nullptr,
                           unsigned* save_vstats = nullptr);
struct QSchedule;
hipError_t launch_als_ne(const AlsArgs& a, const QSchedule& q, bool implicit, double* row_loss, hipStream_t s,
                         hipEvent_t* ev_slot = nullptr);
// global bias + conjugate gradient: out[r][:] = base - gbias * sum_j (c_j - 1) x_j for the n rows `rows` (one workgroup
// per row), slot_of_row[rows[r]] = r
hipError_t launch_gb_row_terms(const AlsArgs& a, const int32_t* rows, int n, float* out, int32_t* slot_of_row, hipStream_t s);
// ev (optional): 7 events, ev[b] before bucket b's kernel, ev[6] after the last one
hipError_t launch_als_cgq(const AlsArgs& a, const QSchedule& q, bool implicit, hipStream_t s, hipEvent_t* ev = nullptr);

// tile capacity (non-zeros per wave tile) the CG kernels are instantiated for
constexpr int kTileNnz = 32;
constexpr int kWavesPerWG = 4;
constexpr int kRowsPerWGShort = 64;  // rows handed to one short-row workgroup
constexpr int kRowsPerWGLong = 8;    // rows handed to one long-row workgroup

constexpr int kCholMaxGrid = 256 * 96;  // Cholesky / NNLS workgroups (grid-stride over rows); many more than slots (3 per CU): the hardware deals them as slots free up

// number of loss partial slots each launcher writes
size_t cg_loss_slots(int n_cols, int n_long);
size_t chol_loss_slots(int n_cols);

// Optional per-kernel timing: when `ev` is non-null the launchers record ev[0] before the first
// kernel, ev[1] between kernels and ev[2] after the last one (all on stream s).
hipError_t launch_als_cg(const AlsArgs& a, bool implicit, hipStream_t s, hipEvent_t* ev = nullptr);
// register-blocked Cholesky (wrmf_chol.hip); rows beyond kCholLongLen non-zeros go to a second launch that sums the
// rank-one updates in two levels (see there); its workgroups' loss slots follow the main launch's
constexpr int kCholLongLen = 4096;
constexpr int kCholLongGrid = 512;
// rows of 1..kCholLrMax non-zeros, implicit feedback, rank 98..128: the low-rank form of the exact solve (wrmf_chol_lr.hip)
constexpr int kCholLrMax = 64;
constexpr int kCholLrGrid = 65536;   // (many more than workgroup slots: the hardware deals them as slots free up, see build_ne_lists)
size_t chol2_loss_slots(int n_cols);
bool chol_lr_supported(const AlsArgs& a, bool implicit);
bool chol_lrx_supported(const AlsArgs& a, bool implicit);
hipError_t launch_als_chol_lrx(const AlsArgs& a, const int32_t* rows, int n_rows, const unsigned* stats, int loss_slot0,
                               hipStream_t s, hipEvent_t* ev_slot);
hipError_t launch_als_chol_lr(const AlsArgs& a, const int32_t* rows, int n_rows, float* M, float* Mt, unsigned* flags,
                              int loss_slot0, hipStream_t s, hipEvent_t* ev_slot = nullptr);
hipError_t launch_als_chol2(const AlsArgs& a, bool implicit, hipStream_t s, hipEvent_t* ev = nullptr);
// rank <= 64: the main launch of the exact solve as one wave per row (wrmf_chol_wave.hip); same rows, same loss slots
bool chol_wave_supported(int k);
hipError_t launch_als_chol_wave(const AlsArgs& a, bool implicit, int grid, int loss_slot0, hipStream_t s, hipEvent_t* ev_slot);
// rank 65..128 (round 6, wrmf_chol_mf.hip): the rows of kCholLrMax + 1 .. kCholMfMax non-zeros as one wave per row -- assembly on
// the matrix cores into the accumulator registers, blocked Cholesky with the trailing updates on v_mfma_f32_32x32x2_f32.
// Needs a.wave_stats (operand scales).  Loss partials [loss_slot0, loss_slot0 + chol_mf_loss_slots(n_rows, implicit))
constexpr int kCholMfMax = 512;       // = kNeMinLen: beyond it the normal-equation launch (rows split across workgroups) keeps the row
constexpr int kCholMfGrid = 256 * 32; // workgroups of one wave, grid-stride over the rows (longest first)
bool chol_mf_supported(int k);
int chol_mf_grid(int n_rows);
int chol_mf_loss_slots(int n_rows, bool implicit);
hipError_t launch_als_chol_mf(const AlsArgs& a, bool implicit, const int32_t* rows, int n_rows, int loss_slot0, hipStream_t s,
                              hipEvent_t* ev_slot);
// rank 128, implicit feedback, conjugate gradient (round 6, wrmf_cg_mf.hip): the rows of kNeMinLen + 1 .. kCgMfMax non-zeros as one
// wave per row -- both normal-equation matrices in the accumulator registers, CG from the tiles.  The rows beyond kCgMfMax stay
// on wrmf_ne.hip (it splits them across workgroups).  Needs a.ne_stats.  Loss partials [loss_slot0, + cg_mf_loss_slots(n_rows))
constexpr int kCgMfMax = 16384;
constexpr int kCgMfGrid = 256 * 16;   // workgroups of four waves (a row each at a time), grid-stride over the rows (longest first)
bool cg_mf_supported(int k, bool implicit);
int cg_mf_grid(int n_rows);
int cg_mf_loss_slots(int n_rows);
hipError_t launch_als_cg_mf(const AlsArgs& a, const int32_t* rows, int n_rows, int loss_slot0, hipStream_t s, hipEvent_t* ev_slot);
hipError_t launch_als_nnls(const AlsArgs& a, bool implicit, hipStream_t s, hipEvent_t* ev = nullptr);
constexpr int kLuGrid = 64;          // workgroups (and loss slots) of the general-solver fallback
constexpr int kFailCap = 1 << 16;    // rows it can take per half-iteration call
hipError_t launch_als_lu_fallback(const AlsArgs& a, bool implicit, size_t loss_slot0, hipStream_t s);
// the failure block: [0] rows sent to the general solver by the current call, [1] of them unresolved, [2], [3] the same
// summed over the earlier calls since rsparse_hip_take_numeric_failures, [4 ..] the current call's row ids
hipError_t launch_fail_roll(int* fails, hipStream_t s);
constexpr int kSumStageBlocks = 256;
hipError_t launch_sum_partials(const double* partials, size_t n, double* out, hipStream_t s, double* tail = nullptr);
hipError_t launch_bias_shift_values(const float* vals, const int32_t* row_idx, const float* X, int k, int bias_row,
                                    int64_t nnz, float* out, hipStream_t s);
hipError_t launch_bias_sweep(const int32_t* p, const int32_t* i, const float* x, const float* other, int n_cols,
                             float lambda, int dynamic_lambda, int non_negative, float* out, hipStream_t s);
hipError_t launch_values_sum(const float* x, int64_t n, double* partials, double* out, hipStream_t s);
hipError_t launch_bias_implicit_terms(const float* vals, const int32_t* row_idx, const float* X, int k, int bias_row,
                                      int64_t nnz, float global_bias, float* rhs_vals, float* loss_tgt, hipStream_t s);
// out[t] = - sum_e X[off + t, e] * ((bias_row >= 0 ? X[bias_row, e] : 0) + global_bias),  t < k1
hipError_t launch_bias_rhs_init(const float* X, int k, int off, int k1, int bias_row, float global_bias, int n,
                                float* scratch, float* out, hipStream_t s);
size_t bias_rhs_init_scratch_floats();
// dst (n x k1p, k1p = k1 rounded up to a multiple of 4) = columns [src_off, src_off + k1) of src (n rows of src_stride floats),
// zeros beyond; Gp (k1p x k1p) = G (k1 x k1) with an identity on the padded diagonal (wrmf_bias.hip: why)
hipError_t launch_pad_rows(const float* src, int src_stride, int src_off, int k1, int k1p, int64_t n, float* dst, hipStream_t s);
hipError_t launch_pad_gramian(const float* G, int k1, int k1p, float* Gp, hipStream_t s);
hipError_t launch_bias_implicit_prep(const int32_t* p, const float* x, int n_cols, int n_other, double lambda,
                                     double* means, double* adj, hipStream_t s);
hipError_t launch_bias_implicit_sweep(const int32_t* p, const int32_t* i, const float* x, const float* other, int n_cols,
                                      int n_other, const double* other_sum, const double* means, const double* adj,
                                      int non_negative, double global_bias, float* out, hipStream_t s);
hipError_t launch_values_subtract_mean(float* x, int64_t n, const double* sum, double inv_count, hipStream_t s);

// Gramian: scratch must hold gramian_scratch_floats(k, n) floats.
size_t gramian_scratch_floats(int k, int64_t n);
hipError_t launch_gramian(const float* X, int k, int64_t n, float ridge, float* XtX, double* sumsq, float* scratch,
                          hipStream_t s, hipEvent_t* ev = nullptr, unsigned* absmax_bits = nullptr);
hipError_t launch_weighted_sumsq(const float* X, int k, int64_t n, const float* w, double* out,
                                 double* scratch /* >= 1024 doubles */, hipStream_t s);
hipError_t launch_f64_to_f32(const double* in, float* out, size_t n, hipStream_t s);
hipError_t check_row_indices_device(const int32_t* i, int64_t nnz, int n_rows, hipStream_t s, int* bad_index);
hipError_t transpose_csc_device(int n_rows, int n_cols, int64_t nnz, const int32_t* p, const int32_t* i, const float* x,
                                int32_t* pt, int32_t* it, float* xt, hipStream_t s, int* bad_index);
hipError_t launch_f32_to_f64(const float* in, double* out, size_t n, hipStream_t s);

// scores = U V^T per user fused with top-k and exclusions (wrmf_topk.hip); U: n_users x k, V: n_items x k,
// both row-major; res / scores: n_users x topk row-major, indices 1-based, INT32_MIN / NaN when fewer than topk
hipError_t launch_top_product(const float* U, const float* V, int n_users, int n_items, int k_rank, int topk,
                              const int32_t* nr_ptr, const int32_t* nr_idx, const int32_t* excl, int n_excl,
                              float glob_mean, int32_t* res, float* scores, hipStream_t s, float* scratch = nullptr);
// `$predict` ordered like the reference's double product (wrmf_topk.hip): kc >= topk candidates per user from the fp32 kernel,
// re-scored in double (U64 / V64, or the fp32 factors widened when they are null), the reference's heap replayed over them;
// scratch: top_product_f64_scratch_words(n_users, kc, topk) 4-byte words; split_scratch as launch_top_product's (for kc)
size_t top_product_f64_scratch_words(int n_users, int kc, int topk);
hipError_t launch_top_product_f64(const float* U32, const float* V32, const double* U64, const double* V64, int n_users,
                                  int n_items, int rank, int topk, int kc, const int32_t* nr_ptr, const int32_t* nr_idx,
                                  const int32_t* excl, int n_excl, double glob_mean, int32_t* res, double* scores, hipStream_t s,
                                  float* scratch, float* split_scratch);
// a call for few users over many items is split over the items (wrmf_topk.hip): floats of scratch it wants
// (2 x entries + n_users), 0 = not split
size_t top_product_scratch_entries(int n_users, int n_items, int topk);
// ... and what launch_top_product wants in all (floats, 0 = none): those lists, or -- many users, a k whose candidate buffers do
// not fit the LDS -- the global candidate buffers of the two-tile kernel (wrmf_topk.hip "GBUF")
size_t top_product_scratch_floats(int n_users, int n_items, int k_rank, int topk);
bool top_product_wants_gbuf(int n_users, int k_rank, int topk);

// device helpers of the multi-GPU context (wrmf_ctx_kernels.hip / wrmf_ctx.cpp)
hipError_t launch_ctx_accumulate(const float* Gpart, const double* sumsq, double* red, int k, hipStream_t s);
hipError_t launch_ctx_put_absmax(const float* absmax, double* red, int k, hipStream_t s);
hipError_t launch_ctx_reduce(const double* all, int ws, int k, float ridge, float* G, double* scal0, float* absmax, hipStream_t s);
hipError_t launch_ctx_sum(const double* v, int n, double* out, hipStream_t s);
hipError_t launch_ctx_add(const double* src, double* dst, hipStream_t s);   // *dst += *src

int padded_rank(int k);  // 32 / 64 / 128, or 0 if unsupported

// ranks 129..256 (wrmf_wide.hip): every variant of the half-iteration on one kernel family, the row's system a packed lower
// triangle in LDS; the Gramian likewise.  launch_als_wide reads the operands of `a` that the rank <= 128 kernels read
// (col_ptrs, row_idx, vals, X, Y, XtX, k, cg_steps, lambda_loss, dynamic_lambda, rhs_vals, loss_tgt, loss_tgt_const, rhs_init,
// gbias, loss_partials [wide_als_grid(n_cols) slots], fail_counter).
bool wide_supported(int k);
int wide_als_grid(int n_cols);
size_t wide_m2_floats_per_wg(int k);            // NNLS scratch per workgroup
size_t wide_gramian_scratch_floats(int k);
hipError_t launch_als_wide(const AlsArgs& a, bool implicit, unsigned solver, float* m2_scratch, float* lu_scratch, hipStream_t s,
                           int n_lo = -1);   // n_lo: rows of at most that many non-zeros belong to another launch
// plain conjugate gradient at these ranks, rows of at most kWideCgMaxLen non-zeros: one wave per row, no k x k system (wrmf_wide_cg.hip)
constexpr int kWideCgMaxLen = 2048;
bool wide_cg_wave_supported(const AlsArgs& a, unsigned solver);
int wide_cg_wave_grid(int n_cols);
int wide_cg_team_grid(int n_cols);
hipError_t launch_wide_cg_wave(const AlsArgs& a, bool implicit, int n_hi, const int32_t* order, int slot0, hipStream_t s);
hipError_t launch_gramian_wide(const float* X, int k, int64_t n, float ridge, float* XtX, double* sumsq, float* scratch,
                               hipStream_t s);

// Measurement harness (rsparse_hip_profile_*): the launcher that records the event `ev_slot` in front of a kernel also
// says which kernel it is about to launch (host function pointer); rsparse_hip_profile_last_names resolves the pointers
// to the names a profiler prints.  No-op when ev_slot is null.
void prof_note(hipEvent_t* ev_slot, const void* kernel_fn);

// Shared by the translation units of the C ABI (wrmf_capi.cpp, wrmf_f64_capi.cpp): the thread-local error text behind
// rsparse_hip_last_error(), and the device block of the exact solvers' failure counters (see launch_fail_roll; nullptr if
// the workspace could not be set up -- the error text then says why).
int capi_fail(int code, const std::string& msg);
int capi_hip_fail(hipError_t e, const char* what);
int* capi_fail_counters();
// RAII for the stateless entry points: takes the failure counts an earlier device-resident call left on the device when the
// stateless call starts (they are not its own) and hands them back to the next reader when it ends
void capi_fail_carry_add(int64_t unresolved, int64_t fallback);
struct StaleFailures {
  int64_t unresolved = 0, fallback = 0;
  StaleFailures();
  ~StaleFailures() { capi_fail_carry_add(unresolved, fallback); }
  StaleFailures(const StaleFailures&) = delete;
  StaleFailures& operator=(const StaleFailures&) = delete;
};

}  // namespace rsparse_hip
