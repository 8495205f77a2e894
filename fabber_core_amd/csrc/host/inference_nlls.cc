// inference_nlls.cc - the "nlls" technique: host driver of the engine's non-linear least squares
// kernel (csrc/vb_nlls_kernel.h). Replaces NLLSInferenceTechnique::Initialize / DoCalculations of
// the reference (inference_nlls.cc:62-214); the per-voxel minimisation, the NLLS precision and
// the result images are computed on the GPU (fabber_nlls_run_host, fabber_vb_postproc_host).
// There is no CPU path.
#include "inference_nlls.h"

#include "host_model.h"
#include "version.h"

#include "../../../include/fabber_vb.h"

#include <cstring>
#include <memory>

using namespace std;

const void *engine_series(FabberRunData &rundata, bool as_matrix, int32_t &data_f64, int &rows, int &cols); // inference_vb.cc
using NEWMAT::Matrix;

struct NLLSInferenceTechnique::EngineStorage
{
    fvb_config cfg;
    Matrix design;
    vector<unsigned char> phi_index;
    vector<Parameter> params;
    bool host_model = false;
    // more than FVB_MAX_PARAMS parameters: the per-parameter entries as a table (fvb_config.params_ext)
    vector<int32_t> wide_transform, wide_type;
    vector<double> wide_zero, wide_post_mean;
    vector<const double *> wide_images;
    fvb_param_table wide;
};

static OptionSpec NLLS_OPTIONS[] = {
    { "vb-init", OPT_BOOL, "Whether NLLS is being run in isolation or as a pre-step for VB", OPT_NONREQ, "" },
    { "lm", OPT_BOOL, "Whether to use LM convergence (default is L)", OPT_NONREQ, "" },
    { "host-model-threads", OPT_INT, "Host threads evaluating a host-side model (0 = as many as the hardware has, at most 16)", OPT_NONREQ, "0" },
    { "host-model", OPT_BOOL, "Evaluate the forward model on the host even if it has a device body (always the case for models from a model library)", OPT_NONREQ, "" },
    { "" },
};

InferenceTechnique *NLLSInferenceTechnique::NewInstance()
{
    return new NLLSInferenceTechnique();
}

NLLSInferenceTechnique::NLLSInferenceTechnique()
    : initialFwdPosterior(NULL)
    , m_vbinit(false)
    , m_lm(false)
    , m_store(new EngineStorage())
{
}

NLLSInferenceTechnique::~NLLSInferenceTechnique()
{
    delete initialFwdPosterior;
    delete m_store;
}

void NLLSInferenceTechnique::GetOptions(vector<OptionSpec> &opts) const
{
    for (int i = 0; NLLS_OPTIONS[i].name != ""; i++)
        opts.push_back(NLLS_OPTIONS[i]);
}

string NLLSInferenceTechnique::GetDescription() const
{
    return "Non-linear least squares inference technique, running on the MI355X engine.";
}

string NLLSInferenceTechnique::GetVersion() const
{
    return fabber_version();
}

void NLLSInferenceTechnique::Initialize(FwdModel *fwd_model, FabberRunData &rundata)
{
    InferenceTechnique::Initialize(fwd_model, rundata);
    LOG << "NLLSInferenceTechnique::Initialising" << endl;
    m_vbinit = rundata.GetBool("vb-init");

    // inference_nlls.cc:68-82. NumParams() is the deprecated count the reference uses here; models
    // that describe themselves through GetParameterDefaults only have m_num_params.
    const int n = m_num_params;
    MVNDist *post = new MVNDist(n);
    MVNDist junk(n);
    m_model->HardcodedInitialDists(junk, *post);
    const string file = rundata.GetStringDefault("fwd-inital-posterior", "modeldefault");
    if (file != "modeldefault")
    {
        LOG << "NLLSInferenceTechnique::File posterior" << endl;
        post->LoadFromMatrix(file);
    }
    delete initialFwdPosterior;
    initialFwdPosterior = post;
    if (initialFwdPosterior->GetSize() != n)
        throw FabberInternalError("NLLSInferenceTechnique: starting estimate has " + stringify(initialFwdPosterior->GetSize())
            + " entries, the model has " + stringify(n) + " parameters");
    m_lm = rundata.GetBool("lm");
    LOG << "NLLSInferenceTechnique::Done initialising" << endl;
}

void NLLSInferenceTechnique::DoCalculations(FabberRunData &rundata)
{
    EngineStorage &st = *m_store;
    fvb_config &cfg = st.cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = FVB_ABI_VERSION;
    const Matrix &coords = rundata.GetVoxelCoords();
    int series_rows = 0, series_cols = 0;
    const void *series = engine_series(rundata, false, cfg.data_f64, series_rows, series_cols);
    cfg.n_voxels = series_cols;
    cfg.n_times = series_rows;
    // the post-processing kernel reads these as for a VB result without noise entries
    cfg.noise = FVB_NOISE_WHITE;
    cfg.n_phis = 0;
    cfg.convergence = FVB_CONV_MAXITS;
    cfg.max_iterations = 1;

    st.params.clear();
    m_model->GetParameters(rundata, st.params);
    const int P = (int)st.params.size();
    if (P > FVB_MAX_PARAMS_EXT)
        throw FabberInternalError("Models with more than " + stringify(FVB_MAX_PARAMS_EXT) + " parameters are not supported by the MI355X engine");
    cfg.n_params = P;
    const bool wide = P > FVB_MAX_PARAMS; // the per-parameter entries as a table (the wave-per-voxel minimiser reads it)
    DeviceModelSpec spec;
    // A model without a device body (any model library written for the reference), or any model when host-model is
    // set, is evaluated on the host - the minimiser's iterations stay on the GPU (fabber_nlls_run_hostmodel_host).
    const bool device_model = m_model->GetDeviceModel(spec) && !rundata.GetBool("host-model");
    if (!device_model)
    {
        spec = DeviceModelSpec();
        spec.model = FVB_MODEL_HOSTJAC;
    }
    st.host_model = !device_model;
    cfg.model = spec.model;
    for (int i = 0; i < 4; i++)
    {
        cfg.model_iopt[i] = spec.iopt[i];
        cfg.model_dopt[i] = spec.dopt[i];
    }
    if (spec.model == FVB_MODEL_LINEAR)
    {
        if (spec.design.Nrows() != cfg.n_times && cfg.n_voxels > 0)
            throw InvalidOptionValue("basis", stringify(spec.design.Nrows()) + " rows",
                "Design matrix length does not match the data (" + stringify(cfg.n_times) + " timepoints)");
        st.design = spec.design;
        cfg.design = st.design.Store();
    }
    if (wide)
    {
        st.wide_transform.assign(P, 0);
        st.wide_type.assign(P, 0);
        st.wide_zero.assign(P, 0.0);
        st.wide_post_mean.assign(P, 0.0);
        st.wide_images.assign(P, (const double *)NULL);
    }
    for (int k = 0; k < P; k++)
    {
        (wide ? st.wide_transform[k] : cfg.transform[k]) = st.params[k].transform->DeviceCode();
        (wide ? st.wide_post_mean[k] : cfg.post_mean[k]) = initialFwdPosterior->means(k + 1); // Fabber space, as it is (inference_nlls.cc:131)
    }
    if (wide)
    {
        st.wide.transform = st.wide_transform.data();
        st.wide.prior_type = st.wide_type.data();
        st.wide.prior_mean = st.wide.prior_var = st.wide.prior_prec = st.wide.post_var = st.wide_zero.data(); // (unused by the minimiser)
        st.wide.post_mean = st.wide_post_mean.data();
        st.wide.image_prior = st.wide_images.data();
        cfg.params_ext = &st.wide;
        if (!device_model)
            throw FabberInternalError("Models with more than " + stringify(FVB_MAX_PARAMS) + " parameters that are evaluated on the host are not supported under method=nlls by the MI355X engine");
    }
    st.phi_index.clear();
    if (!m_masked_tpoints.empty()) // MaskRows (inference_nlls.cc:110,152,216-234)
    {
        st.phi_index.assign(cfg.n_times, 0);
        for (size_t i = 0; i < m_masked_tpoints.size(); i++)
        {
            const int t = m_masked_tpoints[i];
            if (t < 1 || t > cfg.n_times)
                throw InvalidOptionValue("mt" + stringify(i + 1), stringify(t), "Masked timepoint is outside the time series");
            st.phi_index[t - 1] = 255;
        }
        cfg.phi_index = st.phi_index.data();
    }

    const int V = cfg.n_voxels;
    m_result_image.ReSize(fabber_vb_mvn_rows(P), V);
    m_status.assign(V, 0);
    if (V == 0)
        return;

    fvb_nlls nl;
    fabber_nlls_defaults(&nl);
    nl.lm = m_lm ? 1 : 0; // Levenberg by default, Levenberg-Marquardt with --lm (inference_nlls.cc:124-128)
    fvb_outputs out;
    memset(&out, 0, sizeof(out));
    out.mvn = m_result_image.Store();
    out.status = m_status.data();
    vector<int> iterations(V, 0);
    out.iterations = iterations.data();
    LOG << "NLLSInferenceTechnique::Calculations on the MI355X engine, " << V << " voxels x " << cfg.n_times << " timepoints, "
        << (m_lm ? "Levenberg-Marquardt" : "Levenberg") << " damping" << endl;
    const int device = rundata.GetIntDefault("device", 0, 0);
    int rc;
    if (device_model)
        rc = fabber_nlls_run_host(&cfg, &nl, series, &out, device);
    else
    {
        HostModelContext ctx = { this, m_model, &rundata, NULL, &coords, &rundata.GetVoxelSuppData(), cfg.n_times, cfg.n_params, "", {} };
        std::vector<std::unique_ptr<FwdModel> > copies;
        const int nthreads = host_model_instances(ctx, copies, m_model, rundata, m_log);
        LOG << "NLLSInferenceTechnique::The model is evaluated on the host, " << nthreads << " thread(s)" << endl;
        series = engine_series(rundata, true, cfg.data_f64, series_rows, series_cols); // (the models read the Matrix form)
        rc = fabber_nlls_run_hostmodel_host(&cfg, &nl, series, &out, device, &host_model_linearise, &ctx);
        if (rc == -54 && ctx.error != "")
            throw FabberInternalError(ctx.error);
    }
    if (rc != 0)
        throw FabberInternalError(string("MI355X engine failed: ") + fabber_vb_last_error());
    rundata.Progress(V, V);

    // per-voxel failures: what the reference's catch block does (inference_nlls.cc:186-207)
    static const char *reasons[] = { "", "LinearizedFwdModel::ReCentre: Non-finite values found in offset",
        "LinearizedFwdModel::ReCentre: Non-finite values found in jacobian", "", "NEWMAT exception: matrix is singular" };
    int n_bad = 0;
    for (int v = 0; v < V; v++)
    {
        const int code = m_status[v] & 0xff;
        if (code == 0)
            continue;
        const string msg = reasons[code < 5 ? code : 4];
        if (n_bad < 20)
            LOG << "NLLSInferenceTechnique::NEWMAT Exception in this voxel (" << v + 1 << " at " << coords.at0(0, v) << " "
                << coords.at0(1, v) << " " << coords.at0(2, v) << "):\n" << msg << endl;
        n_bad++;
        if (m_halt_bad_voxel)
            throw FabberInternalError(msg);
    }
    if (n_bad)
        LOG << "NLLSInferenceTechnique::Estimates in " << n_bad << " voxels may be unreliable (precision matrix set manually)" << endl;
}

void NLLSInferenceTechnique::SaveResults(FabberRunData &rundata) const
{
    SaveEngineResults(rundata, m_store->cfg, m_store->params, 0, 0, m_store->host_model);
}
