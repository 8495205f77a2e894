/*
 * fabber_vb.h - C ABI of the MI355X voxelwise Variational Bayes engine.
 *
 * This is the thin shim between the host C++ side (the Vb inference technique, the
 * fabber_capi entry points) and the hand-written HIP kernels. Everything the reference does
 * per voxel inside Vb::DoCalculations (inference_vb.cc:360-576) is expressed as ONE call on a
 * plain-old-data problem description plus flat arrays:
 *
 *   reference                                              here
 *   ---------------------------------------------------    ---------------------------------
 *   Vb::SetupPerVoxelDists        inference_vb.cc:144      fvb_config (initial dists) + kernel prologue
 *   Vb::DoCalculationsVoxelwise   inference_vb.cc:415      fabber_vb_run_device / fabber_vb_run_host
 *   WhiteNoiseModel::UpdateTheta  noisemodel_white.cc:275  in-kernel
 *   WhiteNoiseModel::UpdateNoise  noisemodel_white.cc:228  in-kernel
 *   WhiteNoiseModel::CalcFreeEnergy noisemodel_white.cc:365 in-kernel (need_f)
 *   MVNDist inverse / logdet      dist_mvn.cc:197-265      in-kernel LDL^T
 *   LinearizedFwdModel::ReCentre  fwdmodel_linear.cc:126   in-kernel central differences
 *   FwdModel::EvaluateFabber      fwdmodel.cc:365          in-kernel transform + device model body
 *   Prior::ApplyToMVN             priors.cc:108-181        in-kernel (N, I, A prior types)
 *   ConvergenceDetector::Test     convergence.cc:43-378    in-kernel state machines
 *   MVNDist::Save packing         dist_mvn.cc:377-433      kernel output layout (rows x V)
 *   InferenceTechnique::SaveResults inference.cc:112-281   fabber_vb_postproc_device
 *
 * Layout conventions (all arrays are "row x voxel", voxel fastest, exactly the reference's
 * NEWMAT::Matrix(rows, nVoxels) images and the fabber_capi [t][z][y][x] order after masking,
 * rundata_array.cc:100-133):
 *   data      float or double [n_times][n_voxels] (cfg->data_f64)
 *   mvn       double [n_mvn_rows][n_voxels]   n = n_params + n_phis,
 *                                             n_mvn_rows = n(n+1)/2 + n + 1 (dist_mvn.cc:408)
 *   free_energy  double [n_voxels]
 *   f_history    double [f_history_rows][n_voxels]
 *   status / iterations  int32 [n_voxels]
 *
 * No torch types, no C++ types: plain pointers and sizes only.
 */
#ifndef FABBER_VB_H
#define FABBER_VB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVB_MAX_PARAMS 32
#define FVB_MAX_PARAMS_EXT 128 /* with fvb_config.params_ext (the wave-per-voxel kernels; what their LDS holds decides) */
#define FVB_MAX_PHIS 8
#define FVB_MAX_ALPHAS 4 /* AR(1) coefficients: 2 + ar_cross_terms (noisemodel_ar.cc:360-377) */
#define FVB_ABI_VERSION 9

/* Forward models with a device body (fwdmodel_poly.cc:62, fwdmodel_linear.cc:92,
 * examples/fwdmodel_exp.cc:65). FVB_MODEL_HOSTJAC = model only exists as a host plugin;
 * its offset / Jacobian are supplied per re-centre by the host (fabber_vb_run_hostmodel_host). */
enum fvb_model
{
    FVB_MODEL_POLY = 0,
    FVB_MODEL_LINEAR = 1,
    FVB_MODEL_EXP = 2,
    FVB_MODEL_HOSTJAC = 100
};

/* transforms.h:19-23 */
enum fvb_transform
{
    FVB_TRANSFORM_IDENTITY = 0,
    FVB_TRANSFORM_LOG = 1,
    FVB_TRANSFORM_SOFTPLUS = 2,
    FVB_TRANSFORM_FRACTIONAL = 3,
    FVB_TRANSFORM_ABS = 4
};

/* priors.h prior type codes: 'N'/'-' normal, 'I' image, 'A' ARD, 'M','m','P','p' spatial */
enum fvb_prior
{
    FVB_PRIOR_NORMAL = 0,
    FVB_PRIOR_IMAGE = 1,
    FVB_PRIOR_ARD = 2,
    FVB_PRIOR_SPATIAL_M = 3,
    FVB_PRIOR_SPATIAL_m = 4,
    FVB_PRIOR_SPATIAL_P = 5,
    FVB_PRIOR_SPATIAL_p = 6
};

/* setup.cc:50-58 registered convergence detector names */
enum fvb_convergence
{
    FVB_CONV_MAXITS = 0,       /* "maxits"       convergence.cc:43  */
    FVB_CONV_FCHANGE = 1,      /* "pointzeroone" convergence.cc:86  */
    FVB_CONV_FREDUCE = 2,      /* "freduce"      convergence.cc:117 */
    FVB_CONV_TRIALMODE = 3,    /* "trialmode"    convergence.cc:162 */
    FVB_CONV_LM = 4            /* "lm"           convergence.cc:278 */
};

enum fvb_noise
{
    FVB_NOISE_WHITE = 0, /* noisemodel_white.cc */
    FVB_NOISE_AR1 = 1    /* noisemodel_ar.cc: n_phis = num-echoes (1 or 2), fvb_config.ar_cross_terms */
};

/* Per-voxel status word written by the kernel. Mirrors the exceptions the reference's voxel
 * loop catches (inference_vb.cc:529-544). */
enum fvb_status
{
    FVB_OK = 0,
    FVB_BAD_OFFSET = 1,      /* fwdmodel_linear.cc:134-140 non-finite model prediction  */
    FVB_BAD_JACOBIAN = 2,    /* fwdmodel_linear.cc:174-181 non-finite Jacobian          */
    FVB_BAD_FREE_ENERGY = 3, /* noisemodel_white.cc:445-451 non-finite F                */
    FVB_BAD_RESULT = 4,      /* inference_vb.cc:556-570 zero +- identity fallback       */
    FVB_BAD_AR_ALPHA = 5     /* noisemodel_ar.cc:492-499 negative alpha variance        */
};

/* The per-parameter entries of fvb_config for a model with MORE than FVB_MAX_PARAMS parameters (the reference has no
 * limit): [n_params] arrays in the memory space of the configuration's other pointers. The wave-per-voxel kernels take
 * such problems (voxelwise VB under white or AR(1) noise and method=nlls, any built-in model; up to what their LDS holds:
 * n_times, n_params with 8 (4 T + T P + 7 P^2 + ...) bytes <= 160 KB, AR(1): 3 T P), through fabber_vb_run_host / _device,
 * fabber_nlls_run_host / _device and fabber_vb_postproc_*. */
typedef struct fvb_param_table
{
    const int32_t *transform, *prior_type;
    const double *prior_mean, *prior_var, *prior_prec, *post_mean, *post_var;
    const double *const *image_prior; /* [n_params] pointers ([n_voxels] each, or NULL) */
} fvb_param_table;

/* Problem description. Pointers inside are HOST pointers for fabber_vb_run_host and for the
 * oracle, DEVICE pointers for fabber_vb_run_device (documented per field). */
typedef struct fvb_config
{
    int32_t abi_version; /* must be FVB_ABI_VERSION */
    int32_t n_voxels;
    int32_t n_times;
    int32_t n_params; /* P, forward-model parameters */
    int32_t n_phis;   /* white: number of noise-pattern symbols; AR1: num-echoes (interleaved series), 1 or 2 */
    int32_t noise;    /* enum fvb_noise */

    /* ---- forward model ---- */
    int32_t model;          /* enum fvb_model */
    int32_t model_iopt[4];  /* poly: [0]=degree. exp: [0]=num-exps */
    double model_dopt[4];   /* exp: [0]=dt */
    const double *design;   /* linear: [n_times][n_params] row-major, Jacobian of LinearFwdModel
                               (fwdmodel_linear.cc:53-81). Same memory space as data. */

    /* ---- parameters: FwdModel::GetParameters (fwdmodel.cc:210-282) resolved on host ---- */
    int32_t transform[FVB_MAX_PARAMS];
    int32_t prior_type[FVB_MAX_PARAMS];
    double prior_mean[FVB_MAX_PARAMS]; /* Fabber space (fwdmodel.cc:277) */
    double prior_var[FVB_MAX_PARAMS];  /* Fabber space, DistParams::var()  */
    double prior_prec[FVB_MAX_PARAMS]; /* Fabber space, DistParams::prec() = 1/var */
    double post_mean[FVB_MAX_PARAMS];  /* MODEL space initial posterior (fwdmodel.cc:286-305) */
    double post_var[FVB_MAX_PARAMS];   /* MODEL space */
    const double *image_prior[FVB_MAX_PARAMS]; /* [n_voxels] for prior type I, else NULL */

    /* ---- noise model initial distributions (noisemodel_white.cc:127-164) ---- */
    double noise_prior_b[FVB_MAX_PHIS];
    double noise_prior_c[FVB_MAX_PHIS];
    double noise_post_b[FVB_MAX_PHIS];
    double noise_post_c[FVB_MAX_PHIS];
    double locked_noise_stdev; /* <= 0: not locked (noisemodel_white.cc:110,265) */
    const uint8_t *phi_index;  /* [n_times] 0-based phi per timepoint, 255 = masked timepoint
                                  (noisemodel_white.cc:166-226). NULL => all zeros. */

    /* ---- convergence (convergence.cc) ---- */
    int32_t convergence;  /* enum fvb_convergence */
    int32_t max_iterations;
    int32_t max_trials;
    int32_t need_f;       /* m_needF, inference_vb.cc:242 */
    double min_fchange;   /* also max-fchange for LM */

    /* ---- resume (inference_vb.cc:183-216) ---- */
    const double *init_mvn; /* [n_mvn_rows][n_voxels] continue-from-mvn, or NULL */

    /* ---- output control ---- */
    int32_t f_history_rows; /* 0 = do not record (save-free-energy-history) */
    int32_t data_f64;       /* 0: data is float32 (fabber_capi, fabber_capi.h:109); 1: data is float64
                               (the C++ FabberRunData::SetVoxelData(Matrix) route, rundata.cc:924) */

    /* ---- AR(1) noise (noisemodel_ar.cc:322-377) ---- */
    int32_t ar_cross_terms; /* option ar1-cross-terms: 0 "none" (2 alphas), 1 "same" (3), 2 "dual" (4); must be 0
                               with one echo. The noise block of the MVN is (alphas, phi means): 2 +
                               ar_cross_terms + n_phis entries (noisemodel_ar.cc:287-300) */

    /* ---- AR(1): the alpha distributions of noise-initial-prior / noise-initial-posterior
     * (Vb::InitializeNoiseFromParam, inference_vb.cc:132-142 -> Ar1cParams::InputFromMVN, noisemodel_ar.cc:302-316).
     * The Gamma part of those files arrives through noise_prior_b/c and noise_post_b/c like under white noise. ---- */
    int32_t ar_alpha_given;                    /* bit 0: ar_alpha_prior_* hold the prior; bit 1: ar_alpha_post_* the initial
                                                  posterior; a clear bit: HardcodedInitialDists (noisemodel_ar.cc:391-394:
                                                  zero mean, precision 1e-4 I) */
    double ar_alpha_prior_mean[FVB_MAX_ALPHAS];
    double ar_alpha_prior_prec[FVB_MAX_ALPHAS][FVB_MAX_ALPHAS]; /* symmetric; GetPrecisions() of the file's covariance block */
    double ar_alpha_post_mean[FVB_MAX_ALPHAS];
    double ar_alpha_post_cov[FVB_MAX_ALPHAS][FVB_MAX_ALPHAS];   /* symmetric */

    /* ---- more than FVB_MAX_PARAMS parameters ---- */
    const fvb_param_table *params_ext; /* NULL: the fixed arrays above; else n_params may exceed FVB_MAX_PARAMS and every
                                          per-parameter entry is read from this table (the fixed arrays are ignored) */
} fvb_config;

/* the per-parameter entries wherever they are (host code) */
#define FVB_PARAM(cfg, field, k) ((cfg)->params_ext ? (cfg)->params_ext->field[k] : (cfg)->field[k])

/* Result arrays; any pointer may be NULL if that output is not wanted, except mvn. */
typedef struct fvb_outputs
{
    double *mvn;           /* [n_mvn_rows][n_voxels] */
    double *free_energy;   /* [n_voxels] */
    double *f_history;     /* [f_history_rows][n_voxels] */
    int32_t *f_history_len;/* [n_voxels] entries pushed (inference_vb.cc:496-497,553-554) */
    int32_t *status;       /* [n_voxels] enum fvb_status */
    int32_t *iterations;   /* [n_voxels] m_ctx->it at exit */
} fvb_outputs;

/* Post-processing outputs: InferenceTechnique::SaveResults (inference.cc:112-281) and
 * Vb::SaveResults (inference_vb.cc:966-1051). All optional. */
typedef struct fvb_postproc
{
    double *mean;     /* [n_params][n_voxels] model-space means   (inference.cc:139-147) */
    double *var;      /* [n_params][n_voxels] */
    double *std;      /* [n_params][n_voxels] */
    double *zstat;    /* [n_params][n_voxels] */
    double *modelfit; /* [n_times][n_voxels]  (inference.cc:181-243) */
    double *residuals;/* [n_times][n_voxels] */
    double *noise_mean; /* [n_phis][n_voxels] (inference_vb.cc:981-989) */
    double *noise_std;  /* [n_phis][n_voxels] */
} fvb_postproc;

/* Number of rows of the packed MVN image for n = n_params + n_noise_outputs (dist_mvn.cc:408). */
int32_t fabber_vb_mvn_rows(int32_t n);

/* Library / device introspection. */
int32_t fabber_vb_abi_version(void);
int32_t fabber_vb_device_count(void);
const char *fabber_vb_last_error(void);

/* Which kernel a configuration would dispatch to ("lane<exp,4>" / "wave" / ...). */
const char *fabber_vb_kernel_name(const fvb_config *cfg);

/*
 * Run the voxelwise VB loop. All pointers (data, cfg->design, cfg->image_prior[], cfg->phi_index,
 * cfg->init_mvn and every fvb_outputs member) are DEVICE pointers. The launch is asynchronous on
 * `stream` (a hipStream_t, NULL = default stream). Returns 0 or a negative error code; the
 * message is available from fabber_vb_last_error().
 */
int32_t fabber_vb_run_device(const fvb_config *cfg, const void *data, const fvb_outputs *out, void *stream);

/* As fabber_vb_run_device, for callers that already know the number of unmasked timepoints
 * (n_times minus the 255 entries of phi_index): saves the small device-to-host read of
 * phi_index, so the call never synchronises. */
int32_t fabber_vb_run_device_ex(const fvb_config *cfg, const void *data, const fvb_outputs *out, void *stream,
    int32_t n_unmasked);

/* Same with HOST pointers: uploads, runs, downloads, synchronises. `device` = HIP device index. */
int32_t fabber_vb_run_host(const fvb_config *cfg, const void *data, const fvb_outputs *out, int32_t device);

/* What the reference's caller sees of a finished voxelwise run besides the images: the global
 * free energy (sum over the voxels that finished), the iterations executed and the voxels that
 * stopped on a numerical error (inference_vb.cc:529-544). */
typedef struct fvb_summary
{
    double sum_free_energy; /* 0 unless outputs.free_energy was asked for */
    int64_t sum_iterations; /* 0 unless outputs.iterations was asked for */
    int64_t bad_voxels;     /* 0 unless outputs.status was asked for */
} fvb_summary;

/*
 * fabber_vb_run_host over several devices of this node. Voxels are independent
 * (inference_vb.cc:423-571), so the masked-voxel list is cut into n_devices contiguous blocks,
 * block i runs on HIP device devices[i] - one host thread and one stream per block, the block's
 * columns of the caller's [row][voxel] images copied straight in and out - and nothing is
 * exchanged between the blocks; the summary is added up on the host. A device may appear more than
 * once (two blocks then share it on two streams). devices = NULL: every visible device,
 * n_devices ignored. Results are those of fabber_vb_run_host, bit for bit, for every block size
 * that keeps the same kernel (see fabber_vb_kernel_name). summary may be NULL.
 */
int32_t fabber_vb_run_host_multi(const fvb_config *cfg, const void *data, const fvb_outputs *out,
    const int32_t *devices, int32_t n_devices, fvb_summary *summary);

/* Result images from the packed MVN (device pointers, asynchronous on stream). */
int32_t fabber_vb_postproc_device(const fvb_config *cfg, const void *data, const double *mvn,
    const fvb_postproc *pp, void *stream);
int32_t fabber_vb_postproc_host(const fvb_config *cfg, const void *data, const double *mvn,
    const fvb_postproc *pp, int32_t device);

/*
 * Spatial VB (Vb::DoCalculationsSpatial, inference_vb.cc:578-767; SpatialPrior, priors.cc:183-488;
 * Vb::CalcNeighbours, inference_vb.cc:830-964). Selected by method=spatialvb or by any prior of
 * type M, m, P, p. White noise only.
 */
typedef struct fvb_spatial
{
    const int32_t *coords;   /* [3][n_voxels] 0-based grid coordinates (x row, y row, z row), voxels
                                ordered x fastest then y then z (inference_vb.cc:769-793). HOST pointer
                                in every entry point: the neighbour lists and the sweep order are built
                                on the host. */
    int32_t spatial_dims;    /* 0..3, option spatial-dims (default 3) */
    int32_t update_first_iter; /* option update-spatial-prior-on-first-iteration */
    double spatial_speed;    /* option spatial-speed (default -1 = unlimited) */
    double q1, q2;           /* options spatial-q1 (10), spatial-q2 (1) */
    /* Several GPUs (z-slabs, see the stepwise functions below): this process updates local voxels
     * [owned_begin, owned_end) only; the others are ghost copies of the neighbouring slabs'
     * boundary planes. owned_end <= owned_begin (e.g. both 0) = every voxel is owned. */
    int32_t owned_begin, owned_end;
    int32_t n_voxels_global; /* voxel count of the whole volume (0 = n_voxels): the V of h_K, priors.cc:321 */
    const double *locked_centres; /* [n_params][n_voxels] fixed linearisation centres (option locked-linear-from-mvn,
                                inference_vb.cc:171-181): the set-up re-centre uses them (:225-232) and the spatial
                                loop never re-centres (:695-696). NULL = re-centre on the posterior means. Same
                                memory space as data. */
} fvb_spatial;

/* Same conventions as fabber_vb_run_device / fabber_vb_run_host. The iteration count is
 * cfg->max_iterations (the spatial loop always uses the counting detector, inference_vb.cc:599).
 * progress_cb (may be NULL) is called once per iteration with (iteration, max_iterations) as the
 * reference does (inference_vb.cc:610). */
int32_t fabber_vb_run_spatial_device(const fvb_config *cfg, const fvb_spatial *sp, const void *data,
    const fvb_outputs *out, void *stream, void (*progress_cb)(int, int));
int32_t fabber_vb_run_spatial_host(const fvb_config *cfg, const fvb_spatial *sp, const void *data,
    const fvb_outputs *out, int32_t device, void (*progress_cb)(int, int));

/*
 * The same run step by step, for callers that put collectives between the steps (one process per
 * GPU, the volume cut into z-slabs):
 *
 *   fabber_vb_spatial_open(cfg, sp, data, out, stream, &run)   geometry, buffers, SetupPerVoxelDists
 *   for it in 0 .. max_iterations-1:
 *       if (any spatial prior) and (it > 0 or update_first_iter):
 *           fabber_vb_spatial_ak_sums(run, sums)        this slab's (trace term, quadratic term) per parameter
 *           ... all-reduce(SUM) of sums[n_params][2] over the slabs ...
 *           fabber_vb_spatial_set_ak_sums(run, sums)    a_K from the global sums (priors.cc:314-343)
 *       fabber_vb_spatial_sweep(run, it)                first sweep level by level, second sweep
 *       ... halo exchange: fabber_vb_spatial_copy_means() of the owned boundary planes to the
 *           neighbours, of their planes into the ghost voxels ...
 *   fabber_vb_spatial_close(run)                        result images of every local voxel, free
 *
 * Ghost voxels keep the values they were last given. Exchanging the boundary planes once per iteration
 * makes the cuts block-Jacobi (the ghosts BELOW a slab are one iteration old). The reference's order
 * (inference_vb.cc:614, :675) is kept exactly by running the first sweep in level ranges instead:
 *
 *       for each tick: fabber_vb_spatial_sweep_levels(run, it, lo, hi)   first sweep, levels lo <= level < hi
 *                      ... send the top boundary planes to the slab above, receive the ghosts below ...
 *       fabber_vb_spatial_sweep_noise(run, it)                           second sweep
 *
 * where slab r works on the c-th range of levels at tick c + r: what a voxel reads from the slab below has a
 * lower level and was final one tick earlier, what it reads from the slab above has a higher level and is
 * still the previous iteration's - the single-device sweep, voxel for voxel (fabber_core_amd/spatial_mgpu.py).
 * The level of a voxel is x + y + z of its co-ordinates (x + 2 y + 3 z when a prior reads second neighbours),
 * see fabber_vb_spatial_level_weights. With one process and no ghosts fabber_vb_spatial_sweep is exactly
 * fabber_vb_run_spatial_device. All pointers except `data`, `out` are host pointers unless said otherwise.
 */
typedef struct fvb_spatial_run fvb_spatial_run;
int32_t fabber_vb_spatial_open(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out,
    void *stream, fvb_spatial_run **run);
int32_t fabber_vb_spatial_ak_sums(fvb_spatial_run *run, double *sums /* [n_params][2] */);
/* The same sums before the last addition: one (trace term, quadratic term) pair per parameter and SEGMENT of
 * the owned voxels (a z-plane, cut every 4096 voxels). Segments do not depend on the slab decomposition, so
 * adding all slabs' segments in voxel order gives the single-device sums bit for bit. partials = NULL: only
 * *n_segments is returned; else partials [n_segments][n_params][2]. */
int32_t fabber_vb_spatial_ak_segment_sums(fvb_spatial_run *run, double *partials, int32_t *n_segments);
int32_t fabber_vb_spatial_set_ak_sums(fvb_spatial_run *run, const double *sums /* [n_params][2] */);
int32_t fabber_vb_spatial_sweep(fvb_spatial_run *run, int32_t iteration);
/* the two sweeps separately; level_lo / level_hi are LEVEL VALUES (weighted co-ordinate sums, global) */
int32_t fabber_vb_spatial_sweep_levels(fvb_spatial_run *run, int32_t iteration, int64_t level_lo, int64_t level_hi);
int32_t fabber_vb_spatial_sweep_noise(fvb_spatial_run *run, int32_t iteration);
/* weights (wx, wy, wz) of the level function this run sorts its first sweep by */
int32_t fabber_vb_spatial_level_weights(fvb_spatial_run *run, int32_t weights[3]);
/* The free-energy term of the priors of the LAST voxel of the first sweep (the reference reuses one local
 * variable for every voxel's F, inference_vb.cc:612,689,702): with slabs it is the last slab's; get it there
 * (set = 0) and set it everywhere else (set = 1) before the second sweep when F is evaluated. */
int32_t fabber_vb_spatial_fprior(fvb_spatial_run *run, double *value, int32_t set);
/* posterior means [n_params][v_count] and status [v_count] of local voxels [v_begin, v_begin+v_count):
 * run -> buffer (to_device = 0) or buffer -> run (1). Either buffer may be NULL; host or device memory. */
int32_t fabber_vb_spatial_copy_means(fvb_spatial_run *run, int32_t v_begin, int32_t v_count, double *means, int32_t *status,
    int32_t to_device);
int32_t fabber_vb_spatial_close(fvb_spatial_run *run);

/*
 * Voxelwise VB with a forward model that exists only as host code (cfg->model =
 * FVB_MODEL_HOSTJAC): a FwdModel subclass from a model library written for the reference. The
 * model is evaluated where it lives - `linearise` is called once per re-centre with the voxels
 * still running and must fill, for active voxel a (global index voxel_ids[a], Fabber-space means
 * means[a * n_params + i]), lin[a * n_times * (n_params + 1) ...] with g [n_times] followed by
 * J [n_times][n_params]: the model prediction and its Jacobian about those means as
 * LinearizedFwdModel::ReCentre computes them (fwdmodel_linear.cc:126-182). Everything else of the
 * loop runs on the device. cfg->init_mvn (host pointer) must hold the initial posterior
 * (FwdModel::GetInitialPosterior + the initial noise posterior), white noise only. Host pointers
 * throughout; returns 0 or a negative code (-54: the callback returned non-zero).
 */
typedef int32_t (*fvb_linearise_fn)(void *user, int32_t n_active, const int32_t *voxel_ids, const double *means, double *lin);
int32_t fabber_vb_run_hostmodel_host(const fvb_config *cfg, const void *data, const fvb_outputs *out, int32_t device,
    fvb_linearise_fn linearise, void *user);

/*
 * fabber_vb_run_spatial_host on several devices of the node, driven by the calling process: the volume is cut
 * into z-slabs (cuts on plane boundaries, balanced by voxel count; ghost planes each side: one, two with priors
 * of type P / p), slab r on devices[r] (NULL / 0 = every visible device; a device may be listed several times).
 * The first sweep keeps the reference's order across the cuts. First-neighbour priors (types M, m): every slab runs the
 * slab sweep of the one-device run (one launch per slab and iteration, all at once); a voxel whose z+1 neighbour lives on
 * the device above writes its new mean into that neighbour's inbox there (fine-grained peer memory, system-scope stores,
 * self-validating granules), where the lowest plane waits for it as it waits for a workgroup of its own device.
 * Types P and p take part as what the reference codes them to be - priors whose mean reads no neighbour (priors.cc:455) - and
 * need no hand-over at all. A decomposition the slab form does not take (devices that are not peers, no fine-grained
 * memory, a voxel that fails during a first sweep, a non-finite mean next to a type P / p prior, an inbox that never
 * arrives): the slabs step through the SAME global levels as a pipeline, slab r one chunk of levels behind slab r - 1,
 * boundary planes handed upwards after every chunk - the exact form, second-neighbour sums included.
 * Either way the a_K sums are added over the segments of the voxel list in voxel order: the images are those of the
 * one-device run bit for bit. Fewer slabs than devices where the volume has too few planes. Host pointers throughout;
 * models evaluated on the host are not taken (-56); with locked linearisation centres the run uses devices[0] alone.
 */
int32_t fabber_vb_run_spatial_host_multi(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out,
    const int32_t *devices, int32_t n_devices, void (*progress_cb)(int, int));

/*
 * The same in the steps a caller may want to time apart (bench.py --workload c5 --gpus N): open cuts the slabs and puts
 * each slab's part of the problem on its device (`wanted`: which of the optional images - free_energy, status,
 * iterations - the run should keep; only the non-NULL-ness of its members is read); run is ONE complete spatial VB run
 * on the resident data - neighbour tables, numbering, set-up, every iteration, the result images packed on the devices -
 * and may be called again (another run of the same problem); results copies the owned voxels' images to host memory;
 * close gives everything back. -58: the volume has too few planes for two slabs (take fabber_vb_run_spatial_host);
 * -57: locked linearisation centres. slabs: how many slabs there are and which route the last run took.
 */
typedef struct fvb_spatial_multi fvb_spatial_multi;
int32_t fabber_vb_spatial_multi_open(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *wanted,
    const int32_t *devices, int32_t n_devices, fvb_spatial_multi **handle);
int32_t fabber_vb_spatial_multi_run(fvb_spatial_multi *handle, void (*progress_cb)(int, int));
int32_t fabber_vb_spatial_multi_results(fvb_spatial_multi *handle, const fvb_outputs *out);
int32_t fabber_vb_spatial_multi_slabs(fvb_spatial_multi *handle, int32_t *n_slabs, char *route, int32_t route_len);
int32_t fabber_vb_spatial_multi_close(fvb_spatial_multi *handle);

/*
 * Test hook (fault injection; changes nothing unless called): the next fabber_vb_spatial_multi_run / _run_spatial_host_multi
 * of this thread leaves the inboxes of the slab above `pair` unlinked, so that slab's lowest plane waits for means that
 * never arrive, gives up and the run is repeated as the level-chunk pipeline (tests/test_spatial_mgpu.py). -1: off.
 */
void fabber_vb_test_unlink_slab_pair(int32_t pair);

/*
 * Spatial VB (Vb::DoCalculationsSpatial, inference_vb.cc:578-767) with such a model: any FwdModel of a model
 * library under method=spatialvb, as in the reference. The two places of the loop that run the model - the
 * set-up re-centre (:235) and the re-centre that ends every iteration's second sweep (:695) - call `linearise`
 * for the voxels still taking part, about the means the first sweep left; priors, both sweeps, the a_K updates
 * and F run on the device as for the built-in models (the first sweep with one launch per level: the means must
 * be complete on the host before the second sweep starts). Arguments as fabber_vb_run_spatial_host plus the
 * callback; cfg->init_mvn must hold the initial posterior. Device memory: two buffers of
 * n_voxels x n_times x (n_params + 1) doubles (-55 beyond 48 GB each).
 */
int32_t fabber_vb_run_spatial_hostmodel_host(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out,
    int32_t device, fvb_linearise_fn linearise, void *user, void (*progress_cb)(int, int));

/*
 * Non-linear least squares, method=nlls (NLLSInferenceTechnique::DoCalculations,
 * inference_nlls.cc:94-214; cost function / gradient / Gauss-Newton Hessian NLLSCF :223-290). The
 * minimiser is FSL MISCMATHS nonlin (NL_LM), restated in csrc/vb_nlls_kernel.h. Of fvb_config the
 * call reads: n_voxels, n_times, n_params, model*, design, transform[], phi_index (255 = masked
 * timepoint, inference_nlls.cc:110,152), data_f64 and post_mean[] = the starting estimate IN FABBER
 * SPACE (inference_nlls.cc:131: the means of HardcodedInitialDists' posterior as they are).
 * Outputs: mvn [fabber_vb_mvn_rows(n_params)][n_voxels] (parameters only, no noise entries),
 * status, iterations (minimiser iterations), free_energy (if not NULL: the final sum of squares).
 */
typedef struct fvb_nlls
{
    int32_t lm;             /* option lm: 1 = Levenberg-Marquardt (diagonal scaled by 1 + lambda), 0 = Levenberg
                               (lambda added to the diagonal; inference_nlls.cc:124-128) */
    int32_t max_iterations; /* nonlin default 200 */
    double cf_tolerance;    /* fractional change of the cost that counts as converged, 1e-8 */
    double lambda0;         /* initial damping, 0.1 */
    double lambda_max;      /* give up above this damping, 1e20 */
} fvb_nlls;
/* nonlin's defaults */
void fabber_nlls_defaults(fvb_nlls *nl);
/* Pointer conventions as fabber_vb_run_device / fabber_vb_run_host. */
int32_t fabber_nlls_run_device(const fvb_config *cfg, const fvb_nlls *nl, const void *data, const fvb_outputs *out,
    void *stream, int32_t n_unmasked);
int32_t fabber_nlls_run_host(const fvb_config *cfg, const fvb_nlls *nl, const void *data, const fvb_outputs *out,
    int32_t device);
/* method=nlls with a forward model that exists only as host code (cfg->model = FVB_MODEL_HOSTJAC; the reference's
 * NLLS works with any FwdModel, inference_nlls.cc:94-214): the minimiser runs on the device, `linearise` (as for
 * fabber_vb_run_hostmodel_host) is called once per trial point with the voxels still running. Reads of fvb_config
 * what fabber_nlls_run_host reads except model* / design. -54: the callback returned non-zero. */
int32_t fabber_nlls_run_hostmodel_host(const fvb_config *cfg, const fvb_nlls *nl, const void *data, const fvb_outputs *out,
    int32_t device, fvb_linearise_fn linearise, void *user);

/* The engine's work buffers (the re-laid series, the spatial run's state) come from the current device's
 * stream-ordered memory pool, which keeps them between runs (a caller that fits volume after volume allocates
 * once). This returns what the pool holds to the driver; the reference has no counterpart (host memory, freed by
 * the run). */
void fabber_vb_release_cached_memory(void);
/* The same, keeping up to keep_bytes per device for the next run (the C ABI's fabber_destroy does so with 4 GiB when
 * the last handle of the process goes: a client that runs volume after volume pays for its allocations once, and one
 * that is finished holds a bounded amount). */
void fabber_vb_trim_cached_memory(uint64_t keep_bytes);

/* A caller that hands the SAME host buffers (series, result arrays) to fabber_vb_run_host call after call can pin
 * them once: the uploads and downloads of the pipelined entry point are then asynchronous DMA transfers straight
 * from / into the caller's memory instead of staged pageable copies. Pinning costs about as much as one copy of the
 * buffer, so it pays from the second call on. The range must stay allocated until fabber_vb_unpin_host_buffer. */
int32_t fabber_vb_pin_host_buffer(void *ptr, uint64_t bytes);
int32_t fabber_vb_unpin_host_buffer(void *ptr);

/* Force a kernel variant for A/B measurement: 0 = auto, 1 = lane-per-voxel, 2 = wave-per-voxel. */
void fabber_vb_set_variant(int32_t variant);

/* How k'Qk (the residual sum of squares of the linearised model, noisemodel_white.cc:235,252) is
 * obtained, for A/B measurement: 0 (default) = from the streamed moments, replaced by a direct
 * re-summation for the voxels where the moment form has lost precision; 1 = always the direct
 * re-summation (one more model pass per iteration); 2 = moments only, clamped at zero. */
void fabber_vb_set_residual_mode(int32_t mode);
/* mode 0: the direct re-summation is used where k'Qk < tol * (s + 2|d'u| + |d'Ad|); default 1e-10,
 * i.e. the moments value is only kept where it still has >= 6 significant digits */
void fabber_vb_set_residual_tolerance(double tol);

#ifdef __cplusplus
}
#endif

#endif /* FABBER_VB_H */
