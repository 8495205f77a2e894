/*
 * vb_oracle.h - CPU restatement of fabber_core's voxelwise VB loop.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle. The product
 * library (fabber_core_amd/csrc) never includes, links or calls anything in this directory.
 *
 * Pinning status: the reference cannot be compiled in this environment (every hot-path
 * translation unit needs FSL's armawrap/NEWMAT + miscmaths, which are not in /root/reference
 * and not installed), so the oracle is pinned by
 *   (1) the reference's own stored outputs test/outdata_poly/ and test/outdata_linear_vb/
 *       (fixed points of the loop; tests/golden/, see tests/golden/make_golden.py), and
 *   (2) the reference's analytic known-answer tests (test/test_inference.cc,
 *       test/test_vb.cc, test/test_convergence.cc, test/test_priors.cc) re-expressed in tests/.
 * Third-party arithmetic that is not in the reference tree (NEWMAT .i()/LogDeterminant via
 * armawrap->Armadillo->LAPACK; MISCMATHS::digamma) is restated from its published definition:
 * LU with partial pivoting, and an fp64 digamma (recurrence + asymptotic series).
 */
#ifndef VB_ORACLE_H
#define VB_ORACLE_H

#include "../include/fabber_vb.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Optional per-iteration trace (debugging aid for the parity tests). */
typedef struct oracle_trace
{
    double *means;   /* [max_rows][n_params][n_voxels] posterior means after each iteration */
    double *noise_b; /* [max_rows][n_phis][n_voxels] */
    int32_t max_rows;
} oracle_trace;

/* Voxelwise VB, all pointers are host pointers. Voxels [v_begin, v_end) are processed
 * (0-based, end exclusive) so that the bench can time a bounded sample. Returns 0, or the
 * 1-based index of the first voxel whose failure would make the reference rethrow
 * (halt_bad_voxel != 0 mirrors !allow-bad-voxels, inference.cc:93). */
int32_t oracle_vb_run(const fvb_config *cfg, const void *data, const fvb_outputs *out,
    int32_t v_begin, int32_t v_end, int32_t halt_bad_voxel, const oracle_trace *trace);

/* Which algorithm restates NEWMAT's .i(): 0 = LU with partial pivoting (default), 1 = the unpivoted
 * symmetric sweep of the kernels. See inverse() in vb_oracle.cc. Process-wide, not thread-safe. */
void oracle_set_inverse(int32_t mode);

/* InferenceTechnique::SaveResults / Vb::SaveResults images from a packed MVN. */
int32_t oracle_vb_postproc(const fvb_config *cfg, const void *data, const double *mvn,
    const fvb_postproc *pp);

/* Vb::DoCalculationsSpatial (inference_vb.cc:578-767), white noise; host pointers. */
int32_t oracle_vb_run_spatial(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out);
const char *oracle_last_error(void);

/* method=nlls (inference_nlls.cc:94-214), see vb_oracle_nlls.inc; conventions as oracle_vb_run. */
int32_t oracle_nlls_run(const fvb_config *cfg, const fvb_nlls *nl, const void *data, const fvb_outputs *out,
    int32_t v_begin, int32_t v_end, int32_t halt_bad_voxel);
/* Vb::CalcNeighbours (inference_vb.cc:830-964): nn [n_voxels][6], nn2 [n_voxels][30] 1-based ids
 * (0 = none), n2count [n_voxels]. */
int32_t oracle_calc_neighbours(const int32_t *coords, int32_t n_voxels, int32_t spatial_dims, int32_t *nn,
    int32_t *nn2, int32_t *n2count);

/* SpatialPrior::ApplyToMVN (priors.cc:346-488) for ONE parameter on a grid, for unit tests: with the posterior means
 * `means` [n_voxels] of that parameter and the smoothing precision aK, the prior mean and prior precision every voxel
 * gets (in voxel order; nothing is updated in between). type = FVB_PRIOR_SPATIAL_*. */
int32_t oracle_spatial_prior_apply(const int32_t *coords, int32_t n_voxels, int32_t spatial_dims, int32_t type, double mean0,
    double prec0, double aK, const double *means, double *prior_mean, double *prior_prec);

/* Scalar helpers exposed for unit tests. */
double oracle_gammaln(double x);   /* tools.cc:87-98 */
double oracle_digamma(double x);   /* MISCMATHS::digamma restated in fp64 */
double oracle_transform_to_model(int32_t transform, double x);
double oracle_transform_to_fabber(int32_t transform, double x);
double oracle_transform_to_model_var(int32_t transform, double x);
double oracle_transform_to_fabber_var(int32_t transform, double x);

/* Evaluate the forward model in Fabber space for one parameter vector (fwdmodel.cc:365-382).
 * data (n_times floats) is only used by models that look at the voxel data. */
int32_t oracle_evaluate_fabber(const fvb_config *cfg, const double *params, double *result);

/* Drive a convergence detector with a sequence of F values. For each call to Test(F[i]) the
 * outputs receive: done[i] (return value), save[i] (NeedSave after the call), revert[i]
 * (NeedRevert after the call), alpha[i] (LMalpha after the call). Reset() is called first.
 * Returns the number of Test calls made (stops after the first true). */
int32_t oracle_convergence_trace(int32_t conv, int32_t max_iterations, int32_t max_trials,
    double min_fchange, const double *F, int32_t nF, int32_t *done, int32_t *save, int32_t *revert,
    double *alpha, int32_t stop_at_done);

/* Small dense helpers exposed for unit tests: inverse (returns 0 ok, 1 singular) and
 * log|det| with sign of an n x n row-major matrix. */
int32_t oracle_inverse(int32_t n, const double *a, double *inv);
double oracle_logdet(int32_t n, const double *a, int32_t *sign);

#ifdef __cplusplus
}
#endif

#endif
