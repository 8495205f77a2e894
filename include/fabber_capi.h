/*
 * fabber_capi.h - the C ABI language bindings drive fabber through.
 *
 * These are the 15 entry points of the reference's fabber_capi.h:40-279 (same names, argument
 * order and meaning, same error convention: 0 = ok, < 0 = failure with an optional message of at
 * most FABBER_ERR_MAXC - 1 characters copied to err_buf). Implemented in
 * fabber_core_amd/csrc/host/fabber_capi.cc on top of the MI355X voxelwise VB engine; a client
 * that binds the reference's libfabbercore_shared binds this library unchanged
 * (see INTEGRATION.md).
 *
 * Volumes are float arrays in column-major order: x fastest, then y, z, and the 4th dimension
 * (time / parameter row) slowest. Voxels are those with mask != 0, in that scan order.
 */
#ifndef FABBER_CAPI_H
#define FABBER_CAPI_H

#define FABBER_ERR_MAXC 255
#define FABBER_ERR_FATAL -255
#define FABBER_ERR_NEWMAT -254

#ifdef __cplusplus
extern "C" {
#endif

/* reference: fabber_capi.h:40  Create a run context; NULL on failure. */
void *fabber_new(char *err_buf);

/* :51  Load forward models from a shared library exporting get_num_models / get_model_name /
 * get_new_instance_func. */
int fabber_load_models(void *fab, const char *libpath, char *err_buf);

/* :69  Volume extent and mask (nx*ny*nz ints, non-zero = voxel included; must not be NULL). Call
 * before any data is set. */
int fabber_set_extent(void *fab, unsigned int nx, unsigned int ny, unsigned int nz, const int *mask, char *err_buf);

/* :80  Destroy a context (NULL is ignored). Also drops the global component registries. */
void fabber_destroy(void *fab);

/* :92  Set an option; value "" for boolean options. */
int fabber_set_opt(void *fab, const char *key, const char *value, char *err_buf);

/* :109 Set voxel data: nx*ny*nz*data_size floats. The main timeseries is called "data". */
int fabber_set_data(void *fab, const char *name, unsigned int data_size, const float *data, char *err_buf);

/* :121 Size of an output item in the 4th dimension, or < 0 (-1: no such data). */
int fabber_get_data_size(void *fab, const char *name, char *err_buf);

/* :134 Copy an output item into data_buf (nx*ny*nz*size floats, zeros outside the mask). */
int fabber_get_data(void *fab, const char *name, float *data_buf, char *err_buf);

/* :150 Run the inference. log_buf and err_buf are required; progress_cb may be NULL. */
int fabber_dorun(void *fab, unsigned int log_bufsize, char *log_buf, char *err_buf, void (*progress_cb)(int, int));

/* :169 Options of fabber itself (key NULL or ""), of a method (key "method") or of a model
 * (key "model"): first line = description, then name<TAB>description<TAB>type<TAB>optional<TAB>default. */
int fabber_get_options(void *fab, const char *key, const char *value, unsigned int out_bufsize, char *out_buf, char *err_buf);

/* :183 / :197 Newline-separated names of the known models / inference methods. */
int fabber_get_models(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf);
int fabber_get_methods(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf);

/* :212 / :229 / :244 Parameters (names; names + descriptions) and extra outputs of the
 * configured model; call after all options are set. */
int fabber_get_model_params(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf);
int fabber_get_model_param_descs(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf);
int fabber_get_model_outputs(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf);

/* :260 / :277 Evaluate the configured model for one parameter vector (model space). */
int fabber_model_evaluate(void *fab, unsigned int n_params, float *params, unsigned int n_ts, float *indata,
    float *output, char *err_buf);
int fabber_model_evaluate_output(void *fab, unsigned int n_params, float *params, unsigned int n_ts, float *indata,
    const char *output_name, float *output, char *err_buf);

/* ---- not in the reference's header ----
 * The host library keeps the big image buffers of a finished run (up to FVB_HOST_CACHE_BYTES, default 3 GiB) for the
 * next one instead of unmapping ~1 GB in fabber_destroy and faulting it in again in the next fabber_set_data /
 * fabber_dorun. This gives back everything beyond keep_bytes. */
void fabber_amd_trim_host_cache(unsigned long long keep_bytes);

#ifdef __cplusplus
}
#endif

#endif /* FABBER_CAPI_H */
