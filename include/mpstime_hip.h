/*
 * mpstime_hip.h - C ABI of libmpstime_hip.so, the MI355X (gfx950) sweep engine
 * for MPSTime.jl's DMRG-style training sweep.
 *
 * The reference (hugopstackhouse/MPSTime.jl) has no FFI: the seam this library
 * replaces is the Julia method
 *     fitMPS(W::MPS, training_states_meta, testing_states_meta, opts)
 *     src/Training/RealRealHighDimension.jl:587-890
 * selected at :556-560 / :578-582 (where `use_legacy_ITensor` already switches
 * engines).  Each entry point below cites the reference lines it stands in for.
 * INTEGRATION.md shows the Julia `ccall` shim that binds them.
 *
 * Conventions
 *   - plain C, no C++ types or exceptions cross the boundary;
 *   - every call returns 0 (MPST_OK) or a negative mpst_status; the message of
 *     the last failure on a context is available from mpst_last_error();
 *   - sites, classes and samples are 0-based here (the Julia shim converts);
 *   - the caller owns every host buffer, which only has to stay valid for the
 *     duration of the call; the library owns all device memory;
 *   - one caller thread per context; calls block until the device work is done
 *     unless stated otherwise.
 *
 * Data layouts at the boundary
 *   - encoded series  phi[N][T][d]   (d fastest, i.e. a Julia Array of size (d,T,N));
 *     series sorted by class (asserted, RealRealHighDimension.jl:621-625);
 *   - label_idx[N]    0-based class slot, non-decreasing;
 *   - site tensor j   column-major array of size (d, chi[j], chi[j+1]) and, on the
 *     label site, (d, chi[j], chi[j+1], C): element (s,l,r,c) at
 *     s + d*(l + chi[j]*(r + chi[j+1]*c)).  chi[0] = chi[T] = 1.
 *
 * Element types (opts.dtype, RealRealHighDimension.jl:442).  A context has ONE element type - MPST_F64 (default), MPST_F32,
 *   MPST_C128, MPST_C64 - fixed by its first data set (mpst_set_dataset's `dtype`, or mpst_set_dtype before a device-side
 *   encoding): encoded series and site tensors cross the boundary in that type (complex: interleaved (re, im), i.e. Julia's
 *   ComplexF64 / ComplexF32 arrays).  Complex encodings (Fourier, Sahand, Stoudenmire) need a complex type, as in the reference
 *   (RealRealHighDimension.jl:461-466); their training follows the reference's legacy ITensor engine, the only one that
 *   accepts them (src/legacy_itensor/loss_functions.jl:433-640, RealRealLegacyITensor.jl:2-142): phi~ = conj(ps_l) (x) LE (x)
 *   conj(ps_r) (x) RE, yhat = BT * phi~ (no conjugate on BT), loss -log|yhat|^2, grad = -conj(phi~ / yhat).  Whatever the
 *   element type, overlaps, losses, the reduced gradient, the optimiser step, the Gram matrix of the bond tensor and its
 *   eigen-decomposition are fp64; scalar results (losses, spectra, overlaps) are returned as doubles ((re, im) pairs of
 *   doubles for the overlaps of a complex context).
 */
#ifndef MPSTIME_HIP_H
#define MPSTIME_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: mpst_get_info writes 16 entries (1: 12), mpst_set_dtype / mpst_get_info_n added, element types other than Float64
 *    accepted by mpst_set_dataset / mpst_set_mps.  A host compares mpst_version() with the header it was built against. */
#define MPST_ABI_VERSION 2

typedef enum {
    MPST_OK = 0,
    MPST_ERR_INVALID = -1,      /* bad argument / call order (Julia: ArgumentError) */
    MPST_ERR_UNSUPPORTED = -2,  /* option combination the engine lacks (Julia: ErrorException, cf. loss_functions.jl:166-170) */
    MPST_ERR_DEVICE = -3,       /* HIP / RCCL runtime failure */
    MPST_ERR_SVD = -4,          /* bond-tensor decomposition failed / non-finite spectrum; the
                                   class of failure tune() retries on (hyperparameters/tuning.jl:73-86) */
    MPST_ERR_NOMEM = -5
} mpst_status;

enum { MPST_LOSS_KLD = 0, MPST_LOSS_MSE = 1 };   /* src/Structs/options.jl:318-327 */
enum { MPST_OPT_TSGO = 0, MPST_OPT_GD = 1 };     /* src/Structs/options.jl:298-311 */
enum { MPST_F64 = 0, MPST_F32 = 1, MPST_C128 = 2, MPST_C64 = 3 };
enum { MPST_SVD_DEFAULT = 0, MPST_SVD_JACOBI = 1 };  /* opts.svd_alg, RealRealHighDimension.jl:756,798 */
enum { MPST_TRAIN = 0, MPST_TEST = 1 };

/* The MPSOptions fields the sweep consumes
 * (src/Structs/options.jl:106-143; RealRealHighDimension.jl:591-592,606,716-722). */
typedef struct {
    int32_t chi_max;            /* opts.chi_max  (:720) */
    int32_t update_iters;       /* opts.update_iters (:716) */
    int32_t loss;               /* MPST_LOSS_*  <- opts.loss_grad */
    int32_t optimiser;          /* MPST_OPT_*   <- opts.bbopt */
    int32_t rescale_before;     /* opts.rescale[1]  (loss_functions.jl:109) */
    int32_t rescale_after;      /* opts.rescale[2]  (loss_functions.jl:177) */
    int32_t train_classes_separately;  /* TrainSeparate{B} (:606) */
    int32_t svd_alg;            /* MPST_SVD_* */
    int32_t rebuild_caches;     /* 1: redo both full cache rebuilds per sweep as the reference does
                                   (:770,:804); 0: skip them (bit-identical results) */
    int32_t track_cost;         /* opts.track_cost (loss_functions.jl:50-52,80-82,181-184): record the loss before every
                                   optimiser step and at the updated bond tensor; read them with mpst_get_loss_trace */
    double  eta;                /* opts.eta (:719) */
    double  cutoff;             /* opts.cutoff (:722) */
} mpst_options;

/* Per-sweep result of mpst_sweep (RealRealHighDimension.jl:727,806-808). */
typedef struct {
    double  seconds;            /* device time of the sweep (HIP events) */
    int32_t svd_status;         /* 0 or MPST_ERR_SVD */
    int32_t max_chi;            /* largest bond dimension after the sweep */
    int32_t eig_sweeps_total;   /* diagnostic: Jacobi sweeps summed over the bonds of this sweep */
    int32_t eig_fallbacks;      /* diagnostic: bonds of this sweep on which the fast eigensolver's on-device
                                   verification failed and the Jacobi path was used */
} mpst_sweep_stats;

/* Test hook output of mpst_bond_step: gauge-invariant per-bond quantities. */
#define MPST_MAX_SPECTRUM 512
typedef struct {
    double  loss;               /* loss before the (first) step, loss_functions.jl:371 */
    double  grad_norm;          /* ||grad||_F of the first iteration (:79) */
    double  bt_norm;            /* ||bt_new||_F before the rescale (:177) */
    int32_t chi_new;            /* kept bond dimension after truncation */
    int32_t n_spectrum;         /* number of entries valid in spectrum[] */
    int32_t eig_sweeps;
    int32_t reserved;
    double  spectrum[MPST_MAX_SPECTRUM];  /* all singular values of the (rescaled) bond tensor, descending */
} mpst_bond_debug;

int         mpst_version(void);
const char* mpst_last_error(void* ctx);           /* never freed by the caller; NULL ctx -> creation errors */

/* One context per GPU (one process per GPU).  Replaces nothing in the reference:
 * device selection is new. */
int  mpst_create(void** ctx, int device_id);
void mpst_destroy(void* ctx);

/* Batch sharding over GPUs (new design): every rank holds N/G series;
 * the bond gradient + loss are all-reduced with RCCL once per optimiser step.
 * `unique_id` is the 128-byte ncclUniqueId produced on rank 0 and distributed by the host. */
int  mpst_comm_unique_id(uint8_t out_id[128]);
int  mpst_comm_init(void* ctx, const uint8_t unique_id[128], int nranks, int rank);
/* librccl is bound at run time, not linked: MPST_RCCL_LIB if set, else the copy the host process has already loaded (a
 * Python host that imported torch holds torch's own librccl.so.1), else /opt/rocm/lib's - never two copies in one process.
 * Reports "<path> (<how it was found>)" (or the reason it could not be loaded), the library's ncclGetVersion code and the
 * NCCL_VERSION_CODE of the header this library was compiled against; mpst_comm_init refuses another major version. */
int  mpst_comm_library(char* path_out, int32_t path_cap, int32_t* version_out, int32_t* built_against_out);

/* One-shot direct-write all-reduce over xGMI, next to the RCCL path (the 262 KB message is latency-bound).
 * Every rank allocates an inbox in fine-grained device memory and exports it (64-byte hipIpcMemHandle_t); the host
 * gathers the handles of all ranks (torch.distributed all_gather, MPI, ...) and hands the rank-ordered array to
 * mpst_comm_ipc_attach, after which gradient and evaluation sums go through the one-shot kernel (peer writes + flags,
 * fixed rank-order sum: bit-identical replicas).  Call after mpst_set_options / mpst_set_dataset / mpst_set_mps (the slot
 * size follows the gradient buffer) and again if those change the capacity.  RCCL is not required: without
 * mpst_comm_init the pair (export, attach) alone defines the communicator.  mpst_comm_select switches between the two
 * paths when both exist (0 = RCCL, 1 = one-shot).  2..8 ranks (one node). */
int  mpst_comm_ipc_export(void* ctx, int nranks, int rank, uint8_t handle_out[64]);
int  mpst_comm_ipc_attach(void* ctx, const uint8_t* all_handles /* nranks * 64 bytes, rank order */);
int  mpst_comm_select(void* ctx, int oneshot);

/* EncodedTimeSeriesSet -> device (src/Structs/structs.jl:12-33).  `n_global_per_class`
 * may be NULL on a single GPU; with sharding it carries the global class counts that
 * the loss normalisation uses (loss_functions.jl:367,371,423-424). */
int  mpst_set_dataset(void* ctx, int which, const void* phi, const int32_t* label_idx,
                      int64_t N, int32_t T, int32_t d, int32_t C, int32_t dtype,
                      const int64_t* n_global_per_class);
/* opts.dtype ahead of a device-side encoding (mpst_encode_dataset encodes in fp64 and stores in this type; without it a
 * real basis gives MPST_F64 and a complex basis MPST_C128).  Must precede the data sets and the MPS. */
int  mpst_set_dtype(void* ctx, int32_t dtype);

/* Device-side preprocessing + encoding: the raw N x T matrix (row-major, already sorted by
 * class like mpst_set_dataset's input) goes through transform_train_data / transform_test_data
 * (src/utils.jl:161-275) and the Legendre basis (src/Encodings/bases.jl:70-108) on the GPU and becomes data set
 * `which`, instead of uploading the d times larger product states (encode_dataset,
 * src/Encodings/encodings.jl:120-150).  `median`, `iqr` are the
 * RobustSigmoid parameters of the TRAINING set (Normalization.jl fit, utils.jl:171-176) - inputs, or, with
 * fit_sigmoid = 1 on a training set, OUTPUTS fitted on the device (radix sort of the N T values, type-7
 * quantiles).  Training set
 * (is_test = 0): the min / max of the sigmoid-transformed data are fitted on the device and returned in
 * lo / hi.  Test set (is_test = 1): lo / hi are inputs (the training fit); with rescale_out_of_bounds each
 * series is shifted / scaled into [0, 1] (utils.jl:243-266) and `oob_fix` ([N][2], may be NULL) receives the
 * (shift, scale) applied to every series - (0, 1) where nothing was done.  `seconds` (may be NULL) receives
 * the device time of the encoding kernels. */
#define MPST_BASIS_LEGENDRE         0   /* legendre(norm = true),  bases.jl:81-92 */
#define MPST_BASIS_LEGENDRE_NO_NORM 1   /* legendre_no_norm,       bases.jl:108  (MPSOptions default) */
#define MPST_BASIS_FOURIER          2   /* fourier_encode,         bases.jl:23-42 (mpst_encode_values; mean method of complex models) */
/* the remaining closed-form bases: */
#define MPST_BASIS_STOUDENMIRE      3   /* angle_encode, d = 2,    bases.jl:7-21  (complex) */
#define MPST_BASIS_SAHAND           4   /* sahand_encode, even d,  bases.jl:45-68 (complex) */
#define MPST_BASIS_UNIFORM          5   /* uniform_encode,         bases.jl:2-4   (real) */
typedef struct {
    int32_t basis;
    int32_t sigmoid_transform;      /* MPSOptions.sigmoid_transform */
    int32_t minmax;                 /* MPSOptions.minmax */
    int32_t is_test;
    int32_t rescale_out_of_bounds;  /* test sets only */
    int32_t fit_sigmoid;            /* training sets only: fit median / iqr on the device */
    double  median, iqr;            /* in, or out with fit_sigmoid */
    double  lo, hi;                 /* out for the training set, in for a test set */
    double  data_lb, data_ub;       /* MPSOptions.data_bounds */
    double  range_a, range_b;       /* the basis' input range (-1, 1 for Legendre) */
} mpst_encode_opts;
int  mpst_encode_dataset(void* ctx, int which, const double* X, const int32_t* label_idx,
                         int64_t N, int32_t T, int32_t d, int32_t C, mpst_encode_opts* eo,
                         const int64_t* n_global_per_class, double* oob_fix, double* seconds);
/* The same preprocessing + encoding kernels without a data set: X[N][T] in, the encoded states out to the host,
 * phi_out[N][T][d] doubles for the Legendre bases or (re, im) pairs for MPST_BASIS_FOURIER (fourier_encode,
 * bases.jl:23-42: cispi(f x) / sqrt(d), f = 0, 1, -1, 2, -2, ...) - what mpst_impute_model_run takes for a complex model.
 * eo as for mpst_encode_dataset (fits returned in it); oob_fix [N][2] or NULL. */
int  mpst_encode_values(void* ctx, const double* X, int64_t N, int32_t T, int32_t d, mpst_encode_opts* eo, void* phi_out,
                        double* oob_fix, double* seconds);
/* Encoded values of data set `which` back to the host, [N][T][d] in the context's element type
 * (EncodedTimeSeriesSet.timeseries). */
int  mpst_get_encoded(void* ctx, int which, double* phi_out);

int  mpst_set_options(void* ctx, const mpst_options* o);

/* MPS in / out (the W::MPS argument and the TrainedMPS.mps result). `site[j]` points at
 * site tensor j in the boundary layout above; `label_site` is where the "f(x)" index
 * lives (utils.jl:342-354); training requires T-1 on entry (:19-29). */
int  mpst_set_mps(void* ctx, const void* const* site, const int32_t* chi, int32_t T, int32_t label_site);
int  mpst_get_chi(void* ctx, int32_t* chi_out /*[T+1]*/, int32_t* label_site);
int  mpst_get_mps(void* ctx, void* const* site_out /* T buffers sized from mpst_get_chi */);

/* construct_caches, RealRealHighDimension.jl:45-103,631: the left environments of every site left of
 * the label site and the right environments of every site right of it.  With the label on the last
 * site (the state training starts from) this is construct_caches(W, ...; going_left=true). */
int  mpst_build_caches(void* ctx);

/* One full sweep = RealRealHighDimension.jl:727-808. */
int  mpst_sweep(void* ctx, mpst_sweep_stats* out);

/* One sweep of K independent fits of the same shape in ONE launch chain: contexts with the same T, d, C, chi_max, capacity,
 * class counts and loss on one device (hyper-parameter candidates that share the shape - eta, cutoff, the data, the starting MPS
 * may differ -, CV folds, restarts; the reference runs such fits as separate @distributed tasks, hyperparameters/tuning.jl).
 * Every kernel launch carries all K fits, so a sweep costs the ~1200 launches of ONE fit: K separate contexts driven from K host
 * threads reach 2.3x a single fit on one MI355X (the dispatch rate is the limit), one batched chain about 5-6x for K = 8.
 * Results are those of K mpst_sweep calls, bit for bit.  Headline chain only (Float64, d*chi_max <= 128, <= 8192 series, one rank,
 * update_iters = 1, no track_cost / rebuild_caches / profiling): MPST_ERR_UNSUPPORTED otherwise - drive such fits one by one.
 * `out` ([K], may be NULL): seconds = device time of the whole batch.  Errors are reported on ctxs[0]. */
int  mpst_sweep_batch(void* const* ctxs, int32_t K, mpst_sweep_stats* out);
/* The same for fits dealt over SEVERAL devices of a node (what scales there: the batch-sharded sweep repeats its eigensolver on every
 * rank, independent fits share nothing): the contexts are grouped - group[k] is an arbitrary id, NULL = one group per device -, a
 * group must live on one device and hold fits of one shape (<= 64), and every group is advanced by ONE mpst_sweep_batch on its own
 * host thread, concurrently with the others.  No collective, no data crosses devices.  Results are those of K mpst_sweep calls, bit
 * for bit; out[k].seconds is the device time of fit k's group.  The reference's counterpart is the @distributed loop over candidate
 * fits (hyperparameters/hyperopt_utils.jl:201, tuning.jl). */
int  mpst_sweep_batch_multi(void* const* ctxs, int32_t K, const int32_t* group /* [K] or NULL */, mpst_sweep_stats* out);
/* Optional, before the first sweep of a context: it will be advanced in batches of about K fits.  The engine then splits every
 * gradient block of THIS fit into fewer shares (K fits fill the chip together); the share count fixes the order of the partial
 * sums, so mpst_sweep on the same context gives the same bits as mpst_sweep_batch, while a context without the hint differs from
 * it in the last bits of the gradient. */
int  mpst_set_batch_hint(void* ctx, int32_t K);

/* opts.track_cost: the losses the reference prints during the last mpst_sweep, one row per bond in sweep order (backward
 * half-sweep first): update_iters entries "Loss before step i" (loss_functions.jl:50-52 / :80-82) followed by the loss
 * at the updated bond tensor, "Loss at site lid*rid" (:181-184).  out has 2(T-1) * (update_iters + 1) entries. */
int  mpst_get_loss_trace(void* ctx, double* out);

/* One bond update (:733-762 going left, :777-801 going right) - test hook. */
int  mpst_bond_step(void* ctx, int32_t lid, int32_t going_left, mpst_bond_debug* dbg);

/* MSE_loss_acc / MSE_loss_acc_conf, src/summary.jl:33-114.  conf is C*C row-major
 * [truth][prediction] (may be NULL). */
int  mpst_eval(void* ctx, int which, double* mse, double* kld, double* acc, int64_t* conf);

/* classify(mps, states), src/summary.jl:116-136: predicted class slot per series,
 * and optionally the raw overlaps yhat[N][C] (complex context: [N][C][2], (re, im)). */
int  mpst_classify(void* ctx, int which, int32_t* pred /*[N]*/, double* yhat /*[N][C] or NULL*/);

/* Imputation of missing values, batched over the instances of data set `which`:
 * impute_median / impute_mode / impute_ITS of src/Imputation/MPS_methods.jl:198-330, i.e. precondition (:42-99) +
 * impute_at! (:103-177) with get_median_from_rdm / get_mode_from_rdm / get_sample_from_rdm (sampling_utils.jl:98-275),
 * for every instance at once instead of one @distributed task per instance.  The context holds the trained MPS (label
 * index anywhere); instance i is imputed with the class MPS of its label (expand_label_index, utils.jl:356-370).
 *   missing[N][T]   1 where the value is to be imputed (row-major, instances in the order of the data set);
 *                   the encoded values the data set holds at those sites are ignored
 *   grid_x[ngrid]   the candidate values x_k (range(guess_range...; step = dx), imputation.jl:90) and
 *   grid_phi[ngrid][d]  their encoded states (time-independent encodings, :100-106)
 *   o->method       MPST_IMPUTE_MEDIAN (+ weighted median absolute deviation when get_err), MPST_IMPUTE_MODE,
 *                   MPST_IMPUTE_QUANTILE: inverse-transform sampling with the caller's uniform numbers (impute_ITS with
 *                   rejection_threshold = :none; the reference draws them from a MersenneTwister),
 *                   MPST_IMPUTE_MEAN: expectation value (+ standard deviation when get_err), impute_mean :232-265 with
 *                   get_mean_from_rdm, sampling_utils.jl:66-96; the state the chain is re-conditioned on is the encoding
 *                   of the expectation value itself, evaluated on the device with the closed form of the basis (mean_basis:
 *                   MPST_BASIS_LEGENDRE / _LEGENDRE_NO_NORM / _UNIFORM for real models, _FOURIER / _STOUDENMIRE / _SAHAND
 *                   for complex ones; the data-driven bases have no closed form and are not supported),
 *                   MPST_IMPUTE_ITS_REJECT: get_sample_from_rdm with a rejection threshold (:271-296): median and WMAD
 *                   first, then up to max_trials inverse-transform samples, the first one within
 *                   rejection_threshold * WMAD of the median is kept (the last one drawn if none is); err_out = WMAD
 *   o->order        MPST_IMPUTE_FORWARDS / MPST_IMPUTE_BACKWARDS (impute_order, MPS_methods.jl:107-121): the missing
 *                   sites are visited left to right / right to left, each conditioned on the ones already imputed
 *   u[N][T][max_trials]  uniform numbers in [0, 1) for the two sampling methods (max_trials = 1 for QUANTILE), else NULL
 *   x_out[N][T]     imputed value at every missing site (in the encoding's domain; 0 elsewhere), err_out[N][T] the
 *                   uncertainty measure of the method (0 where there is none)
 * chi_max <= 128, d <= 16, real fp64. */
enum { MPST_IMPUTE_MEDIAN = 0, MPST_IMPUTE_MODE = 1, MPST_IMPUTE_QUANTILE = 2, MPST_IMPUTE_MEAN = 3, MPST_IMPUTE_ITS_REJECT = 4 };
enum { MPST_IMPUTE_FORWARDS = 0, MPST_IMPUTE_BACKWARDS = 1 };
typedef struct {
    int32_t method;
    int32_t order;
    int32_t get_err;                /* get_wmad (median) / get_std (mean) */
    int32_t max_trials;             /* ITS_REJECT only (reference default 10) */
    int32_t mean_basis;             /* MEAN only */
    int32_t reserved;
    double  rejection_threshold;    /* ITS_REJECT only */
} mpst_impute_opts;
int  mpst_impute(void* ctx, int which, const uint8_t* missing, const double* grid_x, const double* grid_phi, int32_t ngrid,
                 const mpst_impute_opts* o, const double* u, double* x_out, double* err_out, double* seconds);

/* The same engine on a model handed over in one call instead of through the context's data set and MPS - what an
 * imputation-only caller has (init_imputation_problem, imputation.jl:143-190, takes a trained MPS and test data) - and
 * the only door for the element types the sweep does not train:
 *   dtype    MPST_DTYPE_C64: site tensors, encoded values and grid states are complex (interleaved re, im doubles) - the
 *            reference's Fourier / Sahand bases (bases.jl:23-68), trained through its legacy ITensor path.  Known sites are
 *            projected with the conjugate state (dag(timeseries_enc), MPS_methods.jl:17), densities are |rho phi|^2.
 *   compute  MPST_COMPUTE_F32: the chain contractions (site tensors, environment matrices, boundary vectors) are stored
 *            and multiplied in fp32 (MFMA f32 16x16x4); the density on the grid, its cumulative sums and every selection
 *            stay fp64.  MPST_COMPUTE_F64: everything fp64.
 * site[j]: (s, l, r) column-major, the label site (s, l, r, c), like mpst_set_mps; phi: [N][T][d]; label_idx[N] in [0, C)
 * in any order.  chi_max <= 128, d <= 16.  mean_basis MPST_BASIS_FOURIER / _STOUDENMIRE / _SAHAND for complex models. */
#define MPST_DTYPE_F64   0
#define MPST_DTYPE_C64   1
#define MPST_COMPUTE_F64 0
#define MPST_COMPUTE_F32 1
typedef struct {
    int64_t N;
    int32_t T, d, C, label_site;
    int32_t dtype, compute;
    const void* const* site;
    const int32_t* chi;                 /* [T+1] */
    const void* phi;
    const int32_t* label_idx;
} mpst_impute_model;
int  mpst_impute_model_run(void* ctx, const mpst_impute_model* m, const uint8_t* missing, const double* grid_x, const void* grid_phi,
                           int32_t ngrid, const mpst_impute_opts* o, const double* u, double* x_out, double* err_out, double* seconds);

/* normalize!(W), RealRealHighDimension.jl:852. */
/* Device seconds of the last imputation call split into its two kernels: [0] the environment pass (k_imp_right, MFMA),
 * [1] the sweep with the densities and selections (k_imp_left). */
int  mpst_get_impute_phases(void* ctx, double* seconds_out /*[2]*/);
/* How the last imputation call ran (at most n entries): out[0] = 1 when the grid states were recognised as the Fourier basis
 * (src/Encodings/bases.jl:23-42) on a uniform grid and the conditional densities |rho phi(x)|^2 and their cumulative trapezoid
 * (src/Imputation/sampling_utils.jl:162-199) were evaluated in closed form instead of from the table of grid states;
 * out[1] = 1 when the sweep over the sites (impute_at!, src/Imputation/MPS_methods.jl:103-177) ran for sixteen instances per
 * workgroup (closed-form densities and panels that fit the LDS; otherwise one instance per workgroup - same results). */
int  mpst_get_impute_info(void* ctx, int32_t* out, int32_t n);

int  mpst_normalize(void* ctx);

/* Diagnostics used by bench.py / tests (no reference counterpart). */
int  mpst_selftest_mfma(void* ctx, const double* A /*16xK row-major*/, const double* B /*Kx16*/,
                        int32_t K, double* C_out /*16x16*/);
int  mpst_selftest_eig(void* ctx, const double* G /*n*n symmetric*/, int32_t n, int32_t alg,
                       double* lambda_out /*n*/, double* E_out /*n*n row-major [i][k]*/, int32_t* sweeps);
/* Per-kernel timing with HIP events recorded on the engine's own stream around every launch
 * of the selected kernel classes during mpst_sweep / mpst_bond_step (bit k of kernel_mask):
 * 0 yhat, 1 grad (fused chain: yhat + gradient partials), 2 grad_reduce + update, 3 gram (fused chain: with the optimiser
 * step), 4 eig_tri (default chain: k_eig_trivec, tridiagonalisation + eigenvectors; large-bond path: the whole eigensolver), 5 split, 6 env (fused chain: + back-split + next
 * bond tensor), 7 bt_assemble, 8 all-reduce, 9 eig_vec (only with MPST_EIG_SPLIT=1), 10 eig_fin.
 * mpst_set_profile also resets the accumulators; mpst_get_profile returns the summed device
 * microseconds and the launch count per class (arrays of 16). */
int  mpst_set_profile(void* ctx, uint32_t kernel_mask);
int  mpst_get_profile(void* ctx, double* total_us /*[16]*/, int64_t* count /*[16]*/);
/* which launch chain the context resolved to for its current sizes / options: out[0] fused chain (bond tensors up to
 * 128 x 128, 6 launches per bond), out[1] large-bond path (d*chi_max > 128), out[2] partial gradients per optimiser step
 * of the fused chain, out[3] 64-series chunks of the unfused chain, out[4] capacity bond dimension, out[5] ranks,
 * out[6] sweeps replayed from a hipGraph, out[7] bonds on which the blocked large-bond eigensolver handed over to the
 * library solver, out[8] bonds whose persistent tridiagonalisation gave up waiting for its peers and was redone one
 * launch per step, out[9] bonds whose XCD-local tridiagonalisation found its workgroups on more than one XCD and was
 * redone with the cross-XCD exchange, out[10] the fused chain runs the sliced bond GEMMs (k_yhat_s + k_grad_s: no partial
 * gradients per workgroup), out[11] shares per gradient block of k_grad_s, out[12] the tridiagonalisation and the
 * eigenvectors of a bond run in one launch (k_eig_trivec), out[13] large-bond sweeps that were redone bond by bond because
 * a bond's on-device verification failed, out[14] large bonds: the eigensolver's verdict is read once per sweep instead of
 * once per bond (no host synchronisation inside a sweep), out[15] 0, or 1 + dtype when the element-typed kernels
 * (csrc/mpst_typed.hip) run the sweep */
int  mpst_get_info(void* ctx, int32_t* out /*[16]*/);
/* the same, at most n entries: for callers compiled against another revision of this header.  Entries beyond the 16 of mpst_get_info:
 * out[16] large bonds the randomised subspace eigensolver attempted (real AND complex element types, d*chi_max > 128: top-chi_max
 * singular triplets of the bond matrix by five GEMM half-steps + a (chi_max + 32)-dimensional Rayleigh-Ritz problem, certified on
 * the device against the Gram matrix), out[17] those whose result was accepted - the others were solved by the exact Householder
 * path.  Both counters are as of the last sweep, batch or bond step that RETURNED (they are read at that call's own synchronisation
 * point; the query does not wait for the stream), out[18] the bonds of a sweep run FOUR launches (k_grad_s, k_gram_upd, k_eig_trivec, k_bond_tail: verification +
 * polish of the eigenvectors, back-split, update_caches!, the next bond's tensor and the next bond's overlaps in the last one;
 * Float64, KLD, d*chi_max <= 128, chi_max <= 32, one rank, update_iters = 1, no track_cost), out[19] sweeps / bond steps in which a
 * tail launch's verification failed and the rest ran on the six-launch chain (whose k_eig_fin has the Jacobi fallback). */
int  mpst_get_info_n(void* ctx, int32_t* out, int32_t n);
/* in-kernel phase times (us) of the last eigensolver launch: tridiagonalisation, bisection,
 * tridiagonal eigenvectors, back-transformation, verification+re-orthonormalisation; us[5] = shader
 * cycles (s_memtime) of the tridiagonalisation, for the effective clock */
int  mpst_get_eig_phases(void* ctx, double* us /*[6]*/);
/* in-kernel phase stamps (us since workgroup 0 started; -1: not taken) of the last stamped k_bond_tail launch - the four-launch chain's
 * last launch, which stands for k_eig_fin + update_caches! + the back-split (RealRealHighDimension.jl:107-203) and the next bond's
 * yhat pass (loss_functions.jl:248-262).  Which bond leaves stamps: MPST_TAIL_STAMP="lid,going_left" (default: the middle bond of the
 * backward half-sweep).  us[0..15] workgroup 0, which also hosts a job of the next bond's tensor when the sweep goes on: start,
 * candidate vectors requested, factors requested, bond dimensions known, everything requested, factors in LDS, truncation rule,
 * candidates in LDS, verified + polished, overlap product issued, dense S tile, new environment rows, z + row dot, tile done, role
 * done, stores drained; us[16..31] the same for the last workgroup (a tile, no role); us[48..51] bonds since the context was created
 * by how far from orthonormal their candidate vectors were: |Z^T Z - I| below 1e-13 (nothing to do), below 1e-8 (first-order polish),
 * below 3e-5 (second-order), above (two passes); us[52], us[53], us[54] the earliest start, the latest end and the latest start over
 * ALL workgroups of the stamped launch */
int  mpst_get_tail_phases(void* ctx, double* us /*[55]*/);

#ifdef __cplusplus
}
#endif
#endif /* MPSTIME_HIP_H */
