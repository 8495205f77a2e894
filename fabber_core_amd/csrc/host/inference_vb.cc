// inference_vb.cc - the "vb" technique: host driver of the MI355X voxelwise VB engine.
//
// Replaces Vb::Initialize / DoCalculations / SaveResults of the reference
// (inference_vb.cc:100-130, 360-576, 966-1051) and InferenceTechnique::SaveResults
// (inference.cc:112-281). Everything per-voxel happens inside fabber_vb_run_host /
// fabber_vb_postproc_host (HIP kernels); this file only resolves options into the engine's
// problem block, turns per-voxel status words back into the reference's error behaviour, and
// stores the result images in the run data. There is no CPU path: without a GPU the engine
// call fails and the failure is reported as a FabberInternalError.
#include "inference_vb.h"
#include "tools.h"

#include "convergence.h"
#include "priors.h"
#include "host_model.h"
#include "version.h"

#include "../../../include/fabber_vb.h"

#include <cstring>
#include <math.h>
#include <algorithm>
#include <memory>
#include <thread>

using namespace std;
using NEWMAT::ColumnVector;
using NEWMAT::Matrix;

// ---------------------------------------------------------------------------------------------
// InferenceTechnique base
// ---------------------------------------------------------------------------------------------
std::vector<std::string> InferenceTechnique::GetKnown()
{
    return InferenceTechniqueFactory::GetInstance()->GetNames();
}

InferenceTechnique *InferenceTechnique::NewFromName(const string &name)
{
    InferenceTechnique *inf = InferenceTechniqueFactory::GetInstance()->Create(name);
    if (!inf)
        throw InvalidOptionValue("method", name, "Unrecognized inference method");
    return inf;
}

void InferenceTechnique::UsageFromName(const string &name, std::ostream &stream)
{
    std::unique_ptr<InferenceTechnique> inf(NewFromName(name));
    stream << "Usage information for method: " << name << endl << endl << inf->GetDescription() << endl << endl << "Options: " << endl << endl;
    vector<OptionSpec> options;
    inf->GetOptions(options);
    for (size_t i = 0; i < options.size(); i++)
        stream << options[i];
}

InferenceTechnique::InferenceTechnique()
    : m_model(NULL)
    , m_num_params(0)
    , m_halt_bad_voxel(true)
{
}

InferenceTechnique::~InferenceTechnique()
{
    for (size_t i = 0; i < resultMVNs.size(); i++)
        delete resultMVNs[i];
}

void InferenceTechnique::Initialize(FwdModel *fwd_model, FabberRunData &rundata)
{
    m_log = rundata.GetLogger();
    m_debug = rundata.GetBool("debug");
    m_model = fwd_model;
    vector<Parameter> params;
    m_model->GetParameters(rundata, params);
    m_num_params = (int)params.size();
    LOG << "InferenceTechnique::Model has " << m_num_params << " parameters" << endl;
    m_masked_tpoints = rundata.GetIntList("mt", 1);
    m_halt_bad_voxel = !rundata.GetBool("allow-bad-voxels");
    LOG << (m_halt_bad_voxel ? "InferenceTechnique::Note: numerical errors in voxels will cause the program to halt.\n"
                               "InferenceTechnique::Use --allow-bad-voxels (with caution!) to keep on calculating.\n"
                             : "InferenceTechnique::Using --allow-bad-voxels: numerical errors in a voxel only stop that voxel.\n");
}

void InferenceTechnique::SaveResults(FabberRunData &rundata) const
{
    if (rundata.GetBool("save-mvn"))
    {
        Matrix image;
        image.ReSizeNoInit(m_result_image.Nrows(), m_result_image.Ncols());
        const size_t cols = (size_t)m_result_image.Ncols();
        const double *src = m_result_image.Store();
        double *dst = image.Store();
        fabber_parallel_for(m_result_image.Nrows(), [=](int r) { memcpy(dst + (size_t)r * cols, src + (size_t)r * cols, sizeof(double) * cols); });
        rundata.SaveVoxelDataMove("finalMVN", image, VDT_MVN);
    }
}

// ---------------------------------------------------------------------------------------------
// Vb
// ---------------------------------------------------------------------------------------------
struct Vb::EngineStorage
{
    fvb_config cfg;
    Matrix design;                           // T x P row-major = engine layout
    vector<unsigned char> phi_index;         // per timepoint
    vector<vector<double> > image_priors;    // per parameter
    Matrix init_mvn;                         // continue-from-mvn
    vector<Parameter> params;
    vector<string> model_outputs;
    bool has_device_model;
    // more than FVB_MAX_PARAMS parameters: the per-parameter entries as a table (fvb_config.params_ext)
    vector<int32_t> wide_transform, wide_type;
    vector<double> wide_mean, wide_var, wide_prec, wide_post_mean, wide_post_var;
    vector<const double *> wide_images;
    fvb_param_table wide;
};

static OptionSpec VB_OPTIONS[] = {
    { "noise", OPT_STR, "Noise model to use (white or ar1)", OPT_REQ, "" },
    { "host-model-threads", OPT_INT, "Host threads evaluating a host-side model (0 = as many as the hardware has, at most 16)", OPT_NONREQ, "0" },
    { "host-model", OPT_BOOL, "Evaluate the forward model on the host even if it has a device body (always the case for models from a model library)", OPT_NONREQ, "" },
    { "convergence", OPT_STR, "Name of method for detecting convergence", OPT_NONREQ, "maxits" },
    { "max-iterations", OPT_STR, "number of iterations of VB to use with the maxits convergence detector", OPT_NONREQ, "10" },
    { "min-fchange", OPT_STR, "When using the fchange convergence detector, the change in F to stop at", OPT_NONREQ, "10" },
    { "max-trials", OPT_STR, "When using the trial mode convergence detector, the maximum number of trials after an initial reduction in F", OPT_NONREQ, "10" },
    { "print-free-energy", OPT_BOOL, "Output the free energy in the log", OPT_NONREQ, "" },
    { "save-free-energy-history", OPT_BOOL, "Save the free energy of every iteration", OPT_NONREQ, "" },
    { "mcsteps", OPT_INT, "Number of motion correction steps", OPT_NONREQ, "0" },
    { "continue-from-mvn", OPT_MVN, "Continue previous run from output MVN files", OPT_NONREQ, "" },
    { "output-only", OPT_BOOL, "Skip model fitting, just output requested data based on supplied MVN", OPT_NONREQ, "" },
    { "noise-initial-prior", OPT_MATRIX, "MVN of initial noise prior", OPT_NONREQ, "" },
    { "noise-initial-posterior", OPT_MATRIX, "MVN of initial noise posterior", OPT_NONREQ, "" },
    { "noise-pattern", OPT_STR, "repeating pattern of noise variances for each point (e.g. 12 gives odd and even data points different variances)", OPT_NONREQ, "1" },
    { "PSP_byname<n>", OPT_STR, "Name of model parameter to use prior", OPT_NONREQ, "" },
    { "PSP_byname<n>_type", OPT_STR, "Type of prior to use for parameter <n> - I=image prior", OPT_NONREQ, "" },
    { "PSP_byname<n>_image", OPT_IMAGE, "Image prior for parameter <n>", OPT_NONREQ, "" },
    { "PSP_byname<n>_prec", OPT_FLOAT, "Precision to apply to image prior for parameter <n>", OPT_NONREQ, "" },
    { "PSP_byname<n>_transform", OPT_STR, "Transform to apply to parameter <n>", OPT_NONREQ, "" },
    { "allow-bad-voxels", OPT_BOOL, "Continue if numerical error found in a voxel, rather than stopping", OPT_NONREQ, "" },
    { "ar1-cross-terms", OPT_STR, "For AR1 noise, type of cross-linking (dual, same or none)", OPT_NONREQ, "dual" },
    { "num-echoes", OPT_INT, "For AR1 noise, number of interleaved echoes (1 or 2)", OPT_NONREQ, "1" },
    { "spatial-dims", OPT_INT, "Number of spatial dimensions", OPT_NONREQ, "3" },
    { "spatial-speed", OPT_STR, "Restrict speed of spatial smoothing", OPT_NONREQ, "-1" },
    { "param-spatial-priors", OPT_STR, "Type of spatial priors for each parameter, as a sequence of characters. N=nonspatial, M=Markov random field, P=Penny, A=ARD", OPT_NONREQ, "N+" },
    { "locked-linear-from-mvn", OPT_MVN, "MVN file containing fixed centres for linearization", OPT_NONREQ, "" },
    { "spatial-slabs", OPT_BOOL, "With devices=: cut a spatial VB volume into z-slabs over the listed devices (exact, slower than one device; for volumes beyond one device's memory)", OPT_NONREQ, "" },
    { "" },
};

InferenceTechnique *Vb::NewInstance()
{
    return new Vb();
}

Vb::Vb()
    : m_noise(NULL)
    , m_noise_params(0)
    , m_saveF(false)
    , m_saveFsHistory(false)
    , m_printF(false)
    , m_needF(false)
    , m_locked_linear(false)
    , m_nvoxels(0)
    , m_store(new EngineStorage())
{
}

Vb::~Vb()
{
    delete m_noise;
    delete m_store;
}

void Vb::GetOptions(vector<OptionSpec> &opts) const
{
    for (int i = 0; VB_OPTIONS[i].name != ""; i++)
        opts.push_back(VB_OPTIONS[i]);
}

string Vb::GetDescription() const
{
    return "Variational Bayes inference technique (voxelwise), running on the MI355X engine. See Chappell et al IEEE "
           "Trans Sig Proc 57:1 (2009)";
}
string Vb::GetVersion() const
{
    return fabber_version();
}

void Vb::Initialize(FwdModel *fwd_model, FabberRunData &rundata)
{
    InferenceTechnique::Initialize(fwd_model, rundata);
    m_noise = NoiseModel::NewFromName(rundata.GetString("noise"));
    m_noise->Initialize(rundata);
    m_noise_params = m_noise->NumOutputParams();
    LOG << "Vb::Noise has " << m_noise_params << " parameters" << endl;
    m_saveF = rundata.GetBool("save-free-energy");
    m_saveFsHistory = rundata.GetBool("save-free-energy-history");
    m_printF = rundata.GetBool("print-free-energy");
    rundata.GetStringDefault("mcsteps", "0"); // read but unused, as in the reference
    m_locked_linear = rundata.GetStringDefault("locked-linear-from-mvn", "") != "";
}

bool Vb::IsSpatial(FabberRunData &rundata) const
{
    if (rundata.GetString("method") == "spatialvb")
        return true;
    for (size_t k = 0; k < m_store->params.size(); k++)
    {
        switch (m_store->params[k].prior_type)
        {
        case PRIOR_SPATIAL_M:
        case PRIOR_SPATIAL_m:
        case PRIOR_SPATIAL_P:
        case PRIOR_SPATIAL_p:
            return true;
        }
    }
    return false;
}

// The main series as the engine reads it: float32 where the run data holds it so (every volume set through the C
// ABI, FabberRunData::SetVoxelDataF32), else the double matrix. as_matrix: the caller needs the matrix anyway (a
// model evaluated on the host is handed NEWMAT columns).
const void *engine_series(FabberRunData &rundata, bool as_matrix, int32_t &data_f64, int &rows, int &cols)
{
    const float *f32 = as_matrix ? NULL : rundata.GetMainVoxelDataF32(rows, cols);
    if (f32)
    {
        data_f64 = 0;
        return f32;
    }
    const Matrix &data = rundata.GetMainVoxelData();
    rows = data.Nrows();
    cols = data.Ncols();
    data_f64 = 1; // (rows = timepoints, columns = voxels)
    return data.Store();
}

void Vb::BuildEngineConfig(FabberRunData &rundata, fvb_config &cfg)
{
    EngineStorage &st = *m_store;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = FVB_ABI_VERSION;
    {
        int rows = 0, cols = 0;
        (void)engine_series(rundata, false, cfg.data_f64, rows, cols);
        cfg.n_voxels = cols;
        cfg.n_times = rows;
    }

    // ---- model ----
    st.params.clear();
    m_model->GetParameters(rundata, st.params);
    const int P = (int)st.params.size();
    if (P > FVB_MAX_PARAMS_EXT)
        throw FabberInternalError("Models with more than " + stringify(FVB_MAX_PARAMS_EXT) + " parameters are not supported by the MI355X engine");
    cfg.n_params = P;
    // beyond FVB_MAX_PARAMS the per-parameter entries travel as a table (the wave-per-voxel kernels take such problems:
    // voxelwise VB, white or AR(1) noise, a built-in model - the engine says so where that is not the case)
    const bool wide = P > FVB_MAX_PARAMS;
    if (wide)
    {
        st.wide_transform.assign(P, 0);
        st.wide_type.assign(P, 0);
        st.wide_mean.assign(P, 0);
        st.wide_var.assign(P, 0);
        st.wide_prec.assign(P, 0);
        st.wide_post_mean.assign(P, 0);
        st.wide_post_var.assign(P, 0);
        st.wide_images.assign(P, (const double *)NULL);
    }
    DeviceModelSpec spec;
    // A model without a device body (any model library written for the reference), or any model
    // when host-model is set, is evaluated on the host: 2P+1 Evaluate calls per voxel and
    // re-centre, everything else on the GPU (fabber_vb_run_hostmodel_host).
    st.has_device_model = m_model->GetDeviceModel(spec) && !rundata.GetBool("host-model");
    if (!st.has_device_model)
    {
        spec = DeviceModelSpec();
        spec.model = FVB_MODEL_HOSTJAC;
    }
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

    // ---- parameters / priors ----
    st.image_priors.assign(P, vector<double>());
    for (int k = 0; k < P; k++)
    {
        const Parameter &p = st.params[k];
        (wide ? st.wide_transform[k] : cfg.transform[k]) = p.transform->DeviceCode();
        (wide ? st.wide_type[k] : cfg.prior_type[k]) = Prior::DeviceCode(p.prior_type);
        (wide ? st.wide_mean[k] : cfg.prior_mean[k]) = p.prior.mean();
        (wide ? st.wide_var[k] : cfg.prior_var[k]) = p.prior.var();
        (wide ? st.wide_prec[k] : cfg.prior_prec[k]) = p.prior.prec();
        (wide ? st.wide_post_mean[k] : cfg.post_mean[k]) = p.post.mean();
        (wide ? st.wide_post_var[k] : cfg.post_var[k]) = p.post.var();
        if (p.prior_type == PRIOR_IMAGE)
        {
            const Matrix &img = rundata.GetVoxelData(p.options.find("image")->second);
            if (img.Ncols() != cfg.n_voxels || img.Nrows() < 1)
                throw InvalidOptionValue("image prior", p.name, "Image prior must have one value per voxel");
            st.image_priors[k].resize(cfg.n_voxels);
            for (int v = 0; v < cfg.n_voxels; v++)
                st.image_priors[k][v] = img.at0(0, v);
            (wide ? st.wide_images[k] : cfg.image_prior[k]) = st.image_priors[k].data();
        }
    }

    if (wide)
    {
        st.wide.transform = st.wide_transform.data();
        st.wide.prior_type = st.wide_type.data();
        st.wide.prior_mean = st.wide_mean.data();
        st.wide.prior_var = st.wide_var.data();
        st.wide.prior_prec = st.wide_prec.data();
        st.wide.post_mean = st.wide_post_mean.data();
        st.wide.post_var = st.wide_post_var.data();
        st.wide.image_prior = st.wide_images.data();
        cfg.params_ext = &st.wide;
    }

    // ---- noise ----
    m_noise->ConfigureEngine(cfg, cfg.n_times, st.phi_index);
    cfg.phi_index = st.phi_index.empty() ? NULL : st.phi_index.data();
    // noise-initial-prior / noise-initial-posterior: ONE distribution from a matrix file for every voxel
    // (Vb::InitializeNoiseFromParam, inference_vb.cc:132-142 -> NoiseParams::InputFromMVN); for white noise that is
    // a Gamma per precision from its mean and variance (noisemodel_white.cc:70-79, dist_gamma.cc:29-33)
    for (const char *key : { "noise-initial-prior", "noise-initial-posterior" })
    {
        const string filename = rundata.GetStringDefault(key, "modeldefault");
        if (filename == "modeldefault")
            continue;
        LOG << "VbInferenceTechnique::Loading " << key << " distribution from " << filename << endl;
        MVNDist dist(filename, m_log);
        // AR(1): the alphas first, then the precisions (Ar1cParams::InputFromMVN, noisemodel_ar.cc:302-316)
        const int nAlpha = (cfg.noise == FVB_NOISE_AR1) ? 2 + cfg.ar_cross_terms : 0;
        if (dist.GetSize() != nAlpha + cfg.n_phis)
            throw InvalidOptionValue(key, filename, "The distribution has " + stringify(dist.GetSize()) + " entries, the noise model "
                    + stringify(nAlpha + cfg.n_phis));
        const NEWMAT::SymmetricMatrix &cov = dist.GetCovariance();
        const bool prior = string(key) == "noise-initial-prior";
        if (nAlpha)
        {
            // alpha.CopyFromSubmatrix(mvn, 1, nAlpha, true): the block must be independent of the rest (dist_mvn.cc:157-165)
            for (int i = 1; i <= nAlpha; i++)
                for (int j = nAlpha + 1; j <= nAlpha + cfg.n_phis; j++)
                    if (cov(i, j) != 0)
                        throw FabberRunDataError("Covariance found in part of MVN that should be independent from the rest!");
            NEWMAT::SymmetricMatrix block(cov.SymSubMatrix(1, nAlpha));
            // a distribution over the alphas has a positive definite covariance (the reference would carry its inverse
            // - whatever it is - into UpdateAlpha and the free energy)
            NEWMAT::SymmetricMatrix inverse;
            bool usable = true;
            try
            {
                inverse = block.i();
            }
            catch (...)
            {
                usable = false;
            }
            for (int i = 1; usable && i <= nAlpha; i++)
                usable = block(i, i) > 0 && inverse(i, i) > 0 && inverse(i, i) - inverse(i, i) == 0;
            if (!usable)
                throw InvalidOptionValue(key, filename, "The covariance of the AR(1) coefficients needs to be positive definite");
            cfg.ar_alpha_given |= prior ? 1 : 2;
            for (int i = 0; i < nAlpha; i++)
            {
                (prior ? cfg.ar_alpha_prior_mean : cfg.ar_alpha_post_mean)[i] = dist.means(i + 1);
                for (int j = 0; j < nAlpha; j++)
                {
                    if (prior)
                        cfg.ar_alpha_prior_prec[i][j] = inverse(i + 1, j + 1); // MVNDist::GetPrecisions
                    else
                        cfg.ar_alpha_post_cov[i][j] = block(i + 1, j + 1);
                }
            }
        }
        for (int i = nAlpha + 1; i <= nAlpha + cfg.n_phis; i++) // WhiteParams::InputFromMVN, noisemodel_white.cc:70-79; noisemodel_ar.cc:311-315
            for (int j = i + 1; j <= nAlpha + cfg.n_phis; j++)
                if (cov(i, j) != 0)
                    throw FabberRunDataError("Phis should have zero covariance!");
        for (int i = 0; i < cfg.n_phis; i++)
        {
            // a Gamma distribution has a positive mean and a positive variance (the reference would carry a
            // negative scale or a division by zero into its first update)
            const int q = nAlpha + i + 1;
            if (!(dist.means(q) > 0) || !(cov(q, q) > 0))
                throw InvalidOptionValue(key, filename, "Noise precision " + stringify(i + 1) + " needs a positive mean and a positive variance");
            const double b = cov(q, q) / dist.means(q); // GammaDist::SetMeanVariance
            const double c = dist.means(q) / b;
            (prior ? cfg.noise_prior_b : cfg.noise_post_b)[i] = b;
            (prior ? cfg.noise_prior_c : cfg.noise_post_c)[i] = c;
        }
    }

    // ---- convergence ----
    std::unique_ptr<ConvergenceDetector> conv(
        ConvergenceDetector::NewFromName(rundata.GetStringDefault("convergence", "maxits")));
    conv->Initialize(rundata);
    cfg.convergence = conv->DeviceCode();
    cfg.max_iterations = conv->MaxIterations();
    cfg.max_trials = conv->MaxTrials();
    cfg.min_fchange = conv->MinFChange();
    m_needF = conv->UseF() || m_printF || m_saveF || m_saveFsHistory; // inference_vb.cc:242
    cfg.need_f = m_needF ? 1 : 0;
    // every iteration pushes one value and the end of the voxel one more; trial mode and LM can
    // run more iterations than max-iterations
    cfg.f_history_rows = m_saveFsHistory ? (cfg.max_iterations + 2) * (conv->UseF() ? 12 : 1) : 0;

    // ---- resume ----
    bool continue_from_mvn = true;
    try
    {
        st.init_mvn = rundata.GetVoxelData("continue-from-mvn");
    }
    catch (DataNotFound &)
    {
        continue_from_mvn = false;
    }
    if (continue_from_mvn)
    {
        LOG << "Vb::Continuing from MVN" << endl;
        const int rows = fabber_vb_mvn_rows(P + m_noise_params);
        if (st.init_mvn.Nrows() != rows)
            throw FabberRunDataError("MVNDist::Load  - Incorrect number of rows for an MVN input");
        if (st.init_mvn.Ncols() != cfg.n_voxels)
            throw FabberRunDataError("MVNDist::Load - MVN input has the wrong number of voxels");
        for (int v = 0; v < cfg.n_voxels; v++)
            if (st.init_mvn.at0(rows - 1, v) != 1)
                throw FabberRunDataError("MVNDist::Load - Voxel data does not contain a valid MVN - last value != 1");
        cfg.init_mvn = st.init_mvn.Store();
        rundata.GetStringDefault("continue-from-params", "");
    }
}

// ---- host-evaluated models ----------------------------------------------------------------------

// The initial posterior of every voxel as an MVN image (FwdModel::GetInitialPosterior needs the
// voxel's data, fwdmodel.cc:284-313; noise from the noise model's initial posterior)
void Vb::BuildInitialMvn(FabberRunData &rundata, fvb_config &cfg)
{
    if (cfg.init_mvn)
        return; // continue-from-mvn
    // noise entries of the MVN: white = the precisions; AR(1) = the alphas, then the precisions (noisemodel_ar.cc:287-300)
    const int NA = (cfg.noise == FVB_NOISE_AR1) ? 2 + cfg.ar_cross_terms : 0;
    const int P = cfg.n_params, N = cfg.n_phis, n = P + NA + N, V = cfg.n_voxels;
    const int rows = fabber_vb_mvn_rows(n), nCov = n * (n + 1) / 2;
    const Matrix &data = rundata.GetMainVoxelData();
    const Matrix &coords = rundata.GetVoxelCoords();
    const Matrix &supp = rundata.GetVoxelSuppData();
    Matrix &img = m_store->init_mvn;
    img.ReSize(rows, V);
    for (int r = 0; r < rows; r++)
        for (int v = 0; v < V; v++)
            img.at0(r, v) = 0;
    for (int v = 0; v < V; v++)
    {
        if (supp.Ncols() == V)
            m_model->PassData(v + 1, ColumnVector(data.Column(v + 1)), ColumnVector(coords.Column(v + 1)), ColumnVector(supp.Column(v + 1)));
        else
            m_model->PassData(v + 1, ColumnVector(data.Column(v + 1)), ColumnVector(coords.Column(v + 1)));
        MVNDist post(P);
        m_model->GetInitialPosterior(post, rundata);
        const NEWMAT::SymmetricMatrix &cov = post.GetCovariance();
        for (int i = 0; i < P; i++)
        {
            for (int j = 0; j <= i; j++)
                img.at0(i * (i + 1) / 2 + j, v) = cov(i + 1, j + 1);
            img.at0(nCov + i, v) = post.means(i + 1);
        }
        for (int a = 0; a < NA; a++) // Ar1cNoiseModel's initial alpha posterior: N(0, 1e4 I) (noisemodel_ar.cc:379-403),
        {                            // or the one of noise-initial-posterior
            const int q = P + a;
            if (cfg.ar_alpha_given & 2)
            {
                for (int a2 = 0; a2 <= a; a2++)
                    img.at0(q * (q + 1) / 2 + P + a2, v) = cfg.ar_alpha_post_cov[a][a2];
                img.at0(nCov + q, v) = cfg.ar_alpha_post_mean[a];
            }
            else
                img.at0(q * (q + 1) / 2 + q, v) = 1e4;
        }
        for (int k = 0; k < N; k++)
        {
            const double b = cfg.noise_post_b[k], c = cfg.noise_post_c[k];
            const int q = P + NA + k;
            img.at0(q * (q + 1) / 2 + q, v) = b * b * c; // GammaDist variance / mean, as OutputAsMVN
            img.at0(nCov + q, v) = b * c;
        }
        img.at0(rows - 1, v) = 1;
    }
    cfg.init_mvn = img.Store();
}

// one model instance per host thread (host-model-threads, default: the hardware's, at most 16): a FwdModel holds the
// current voxel's data, so instances cannot be shared
int host_model_instances(HostModelContext &ctx, std::vector<std::unique_ptr<FwdModel> > &copies, FwdModel *model, FabberRunData &rundata,
    EasyLog *log)
{
    int nthreads = rundata.GetIntDefault("host-model-threads", 0, 0, 256);
    if (nthreads == 0)
        nthreads = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
    ctx.models.push_back(model);
    for (int k = 1; k < nthreads; k++)
    {
        copies.emplace_back(FwdModel::NewFromName(rundata.GetString("model")));
        copies.back()->SetLogger(log);
        copies.back()->Initialize(rundata);
        vector<Parameter> tmp;
        copies.back()->GetParameters(rundata, tmp); // resolves the transforms EvaluateFabber applies
        ctx.models.push_back(copies.back().get());
    }
    ctx.data = &rundata.GetMainVoxelData();
    return nthreads;
}

// LinearizedFwdModel::ReCentre (fwdmodel_linear.cc:126-182) for active voxels [a0, a1) with one model instance
static void linearise_range(HostModelContext &cx, FwdModel *model, int a0, int a1, const int32_t *ids, const double *means, double *lin)
{
    const int T = cx.T, P = cx.P;
    const bool have_supp = cx.suppdata->Ncols() == cx.data->Ncols();
    ColumnVector centre(P), pert(P), g, f2, f3;
    for (int a = a0; a < a1; a++)
    {
        const int v = ids[a];
        if (have_supp)
            model->PassData(v + 1, ColumnVector(cx.data->Column(v + 1)), ColumnVector(cx.coords->Column(v + 1)),
                ColumnVector(cx.suppdata->Column(v + 1)));
        else
            model->PassData(v + 1, ColumnVector(cx.data->Column(v + 1)), ColumnVector(cx.coords->Column(v + 1)));
        for (int i = 0; i < P; i++)
            centre(i + 1) = means[(size_t)a * P + i];
        double *out = lin + (size_t)a * T * (P + 1);
        model->EvaluateFabber(centre, g, "");
        if (g.Nrows() != T)
            throw FabberInternalError("Model returned " + stringify(g.Nrows()) + " timepoints, data has " + stringify(T));
        for (int t = 0; t < T; t++)
            out[t] = g(t + 1);
        for (int i = 0; i < P; i++)
        {
            double delta = centre(i + 1) * 1e-5; // :157-161
            if (delta < 0)
                delta = -delta;
            if (delta < 1e-10)
                delta = 1e-10;
            pert = centre;
            pert(i + 1) = centre(i + 1) + delta;
            const double c2 = pert(i + 1);
            model->EvaluateFabber(pert, f2, "");
            pert(i + 1) = centre(i + 1) - delta;
            const double c3 = pert(i + 1);
            model->EvaluateFabber(pert, f3, "");
            for (int t = 0; t < T; t++)
                out[T + (size_t)t * P + i] = (f2(t + 1) - f3(t + 1)) / (c2 - c3);
        }
    }
}

// fvb_linearise_fn: the active voxels shared out over the host threads, one model instance each
// (a FwdModel holds the current voxel's data, so instances cannot be shared)
int32_t Vb::LineariseCallback(void *user, int32_t n_active, const int32_t *ids, const double *means, double *lin)
{
    return host_model_linearise(user, n_active, ids, means, lin);
}

int32_t host_model_linearise(void *user, int32_t n_active, const int32_t *ids, const double *means, double *lin)
{
    HostModelContext &cx = *static_cast<HostModelContext *>(user);
    const int nthreads = std::max(1, std::min((int)cx.models.size(), n_active / 64 + 1));
    std::vector<std::string> errors(nthreads);
    auto work = [&](int k) {
        try
        {
            const int a0 = (int)((long long)n_active * k / nthreads), a1 = (int)((long long)n_active * (k + 1) / nthreads);
            linearise_range(cx, cx.models[k], a0, a1, ids, means, lin);
        }
        catch (std::exception &e)
        {
            errors[k] = e.what();
        }
        catch (...)
        {
            errors[k] = "unknown exception in the model";
        }
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < nthreads; k++)
        pool.emplace_back(work, k);
    work(0);
    for (size_t k = 0; k < pool.size(); k++)
        pool[k].join();
    for (int k = 0; k < nthreads; k++)
        if (errors[k] != "")
        {
            cx.error = errors[k];
            return 1;
        }
    return 0;
}

// The engine's per-iteration callback carries no context pointer: the run data whose
// ProgressCheck should hear about spatial iterations (inference_vb.cc:610) is kept per thread.
static thread_local FabberRunData *s_progress_rundata = NULL;
static void spatial_progress(int it, int maxits)
{
    if (s_progress_rundata)
        s_progress_rundata->Progress(it, maxits);
}

void Vb::DoCalculations(FabberRunData &rundata)
{
    FabberStageTimer timer("Vb::DoCalculations");
    fvb_config &cfg = m_store->cfg;
    BuildEngineConfig(rundata, cfg);
    timer.lap("engine configuration");
    m_nvoxels = cfg.n_voxels;
    const int rows = fabber_vb_mvn_rows(cfg.n_params + m_noise_params);
    m_result_image.ReSizeNoInit(rows, m_nvoxels); // (the engine writes every voxel's column)
    m_free_energy.clear();
    m_status.assign(m_nvoxels, 0);
    timer.lap("result image");

    const bool output_only = rundata.GetBool("output-only");
    const bool spatial = IsSpatial(rundata);
    if (m_nvoxels == 0)
        return;

    if (output_only)
    {
        LOG << "Vb::DoCalculations output-only set - not performing any calculations" << endl;
        if (!cfg.init_mvn)
            throw FabberRunDataError("output-only needs continue-from-mvn");
        m_result_image = m_store->init_mvn;
        m_needF = false;
        return;
    }

    // (a model evaluated on the host takes its data as NEWMAT columns: the matrix then, for the engine too)
    int series_rows = 0, series_cols = 0;
    const void *series = engine_series(rundata, !m_store->has_device_model, cfg.data_f64, series_rows, series_cols);
    // (per-voxel arrays the engine writes in full are sized without being filled, and come from the block cache; the free
    // energy - 9999 where the reference stores none, inference_vb.cc:165 - is asked for only when somebody reads it)
    std::vector<int, NEWMAT::DefaultInitAllocator<int> > iterations((size_t)m_nvoxels);
    vector<int> hist_len;
    if (m_needF)
        m_free_energy.assign(m_nvoxels, 9999);
    else
        m_free_energy.clear();
    if (cfg.f_history_rows > 0)
    {
        m_f_history.ReSize(cfg.f_history_rows, m_nvoxels);
        hist_len.assign(m_nvoxels, 0);
    }
    fvb_outputs out;
    memset(&out, 0, sizeof(out));
    out.mvn = m_result_image.Store();
    out.free_energy = m_needF ? m_free_energy.data() : NULL;
    out.status = m_status.data();
    out.iterations = iterations.data();
    if (cfg.f_history_rows > 0)
    {
        out.f_history = m_f_history.Store();
        out.f_history_len = hist_len.data();
    }
    // devices=all | devices=0,1,...: voxelwise VB shards the voxel list over several GPUs of the node
    // (fabber_vb_run_host_multi); everything else runs on the first of them (or on device=<index>)
    vector<int32_t> device_list;
    const string devices_opt = rundata.GetStringDefault("devices", "");
    if (devices_opt != "" && devices_opt != "all")
    {
        string item;
        for (size_t i = 0; i <= devices_opt.size(); i++)
        {
            if (i == devices_opt.size() || devices_opt[i] == ',')
            {
                if (item == "")
                    throw InvalidOptionValue("devices", devices_opt, "Must be 'all' or a comma-separated list of device indices");
                device_list.push_back(convertTo<int>(item));
                item = "";
            }
            else
                item += devices_opt[i];
        }
    }
    const int device = device_list.empty() ? rundata.GetIntDefault("device", 0, 0) : device_list[0];
    const Matrix &coords = rundata.GetVoxelCoords();
    int rc;
    // a model evaluated on the host: one model instance per host thread (host-model-threads, default: the
    // hardware's, at most 16)
    HostModelContext ctx = { this, m_model, &rundata, NULL, &coords, &rundata.GetVoxelSuppData(), cfg.n_times, cfg.n_params, "", {} };
    std::vector<std::unique_ptr<FwdModel> > copies;
    auto prepare_host_model = [&]() {
        const int nthreads = host_model_instances(ctx, copies, m_model, rundata, m_log);
        LOG << "Vb::Model evaluations on " << nthreads << " host thread(s)" << endl;
        series = engine_series(rundata, true, cfg.data_f64, series_rows, series_cols);
    };
    if (m_locked_linear)
    {
        // In the voxelwise loop the locked centres only serve the set-up re-centre, which the loop
        // repeats about the posterior means before anything uses it (inference_vb.cc:227-235, :443,
        // :490): no effect on the results. The spatial loop really keeps them (:695).
        if (!spatial)
            LOG << "Vb::locked-linear-from-mvn has no effect on voxelwise VB (the loop re-centres about the posterior means)" << endl;
    }
    vector<double> locked_centres; // [P][V]
    if (m_locked_linear && spatial)
    {
        // inference_vb.cc:171-181,225-232: the means of the first P entries of the given MVN image ("does not check
        // if the correct number of parameters is present"); MVNDist::Load's checks (dist_mvn.cc:324-374)
        const string file = rundata.GetString("locked-linear-from-mvn");
        LOG << "Vb::Loading fixed linearization centres from the MVN '" << file << "'" << endl;
        const Matrix &mvn = rundata.GetVoxelData(file);
        if (mvn.Ncols() == 0)
            throw FabberRunDataError("MVNDist::Load - Voxel data is empty");
        const int n = ((int)sqrt(double(8 * mvn.Nrows() + 1)) - 3) / 2;
        if (mvn.Nrows() != n * (n + 1) / 2 + n + 1)
            throw FabberRunDataError("MVNDist::Load  - Incorrect number of rows for an MVN input");
        if (n < cfg.n_params || mvn.Ncols() != m_nvoxels)
            throw FabberRunDataError("locked-linear-from-mvn: the MVN does not hold the model's parameters for every voxel");
        const int nCov = n * (n + 1) / 2;
        locked_centres.resize((size_t)cfg.n_params * m_nvoxels);
        for (int v = 0; v < m_nvoxels; v++)
        {
            if (mvn.at0(nCov + n, v) != 1)
                throw FabberRunDataError("MVNDist::Load - Voxel data does not contain a valid MVN - last value != 1");
            for (int k = 0; k < cfg.n_params; k++)
                locked_centres[(size_t)k * m_nvoxels + v] = mvn.at0(nCov + k, v);
        }
    }
    if (spatial)
    {
        // Vb::DoCalculationsSpatial (inference_vb.cc:578-767): whole-volume sweeps with the counting
        // detector; options as SpatialPrior reads them (priors.cc:190-211)
        fvb_spatial sp;
        memset(&sp, 0, sizeof(sp));
        sp.spatial_dims = rundata.GetIntDefault("spatial-dims", 3, 0, 3);
        if (sp.spatial_dims == 1)
            WARN_ONCE("spatial-dims=1 is weird... hope you're just testing!");
        else if (sp.spatial_dims == 2)
            WARN_ONCE("spatial-dims=2 may not work the way you expect");
        sp.spatial_speed = rundata.GetDoubleDefault("spatial-speed", -1);
        sp.update_first_iter = rundata.GetBool("update-spatial-prior-on-first-iteration") ? 1 : 0;
        sp.q1 = rundata.GetDoubleDefault("spatial-q1", 10.0);
        sp.q2 = rundata.GetDoubleDefault("spatial-q2", 1.0);
        if (coords.Nrows() < 3 || coords.Ncols() != m_nvoxels)
            throw FabberInternalError("Vb::CalcNeighbours: voxel co-ordinates do not match the data");
        vector<int32_t> grid((size_t)3 * m_nvoxels);
        for (int d = 0; d < 3; d++)
            for (int v = 0; v < m_nvoxels; v++)
                grid[(size_t)d * m_nvoxels + v] = (int32_t)coords.at0(d, v);
        sp.coords = grid.data();
        sp.locked_centres = locked_centres.empty() ? NULL : locked_centres.data();
        cfg.convergence = FVB_CONV_MAXITS;
        cfg.max_iterations = convertTo<int>(rundata.GetStringDefault("max-iterations", "10"));
        cfg.f_history_rows = 0;
        out.f_history = NULL;
        out.f_history_len = NULL;
        LOG << "Vb::Spatial calculations on the MI355X engine, " << m_nvoxels << " voxels x " << cfg.n_times
            << " timepoints, " << cfg.max_iterations << " iterations" << endl;
        s_progress_rundata = &rundata;
        // devices= shards VOXELWISE VB. Spatial VB over several devices (z-slabs, pipelined first sweep) is exact but
        // slower than one device - the ordered sweep is a latency chain that more devices do not shorten - so it
        // is an explicit choice (spatial-slabs) for volumes beyond one device's memory.
        const bool slabs = rundata.GetBool("spatial-slabs");
        if (devices_opt != "" && !slabs)
            WARN_ONCE("devices= shards voxelwise VB only: this spatial VB run uses one device (the faster choice; "
                      "--spatial-slabs cuts the volume into z-slabs over the listed devices for volumes beyond one device's memory)");
        if (m_store->has_device_model && devices_opt != "" && slabs && !sp.locked_centres)
        {
            // devices=all | devices=0,1,...: z-slabs of the volume on several GPUs, pipelined first sweep
            LOG << "Vb::devices=" << devices_opt << ": the volume is cut into z-slabs" << endl;
            rc = fabber_vb_run_spatial_host_multi(&cfg, &sp, series, &out, device_list.empty() ? NULL : device_list.data(),
                (int32_t)device_list.size(), spatial_progress);
        }
        else if (m_store->has_device_model)
            rc = fabber_vb_run_spatial_host(&cfg, &sp, series, &out, device, spatial_progress);
        if (m_store->has_device_model && rc == -40)
        {
            // no spatial kernels were built for this model with this many parameters: the model's own host code
            // does the re-centres instead (up to 8 parameters)
            LOG << "Vb::no device kernels for spatial VB with " << cfg.n_params << " parameters of this model" << endl;
            m_store->has_device_model = false;
            cfg.model = FVB_MODEL_HOSTJAC;
            cfg.design = NULL;
        }
        if (!m_store->has_device_model)
        {
            // any FwdModel under spatial VB, as in the reference: the model's two re-centres per iteration
            // run here on the host, the sweeps on the device (fabber_vb_run_spatial_hostmodel_host)
            LOG << "Vb::the model is evaluated on the host" << endl;
            BuildInitialMvn(rundata, cfg);
            prepare_host_model();
            rc = fabber_vb_run_spatial_hostmodel_host(&cfg, &sp, series, &out, device, &Vb::LineariseCallback, &ctx, spatial_progress);
            if (rc == -54 && ctx.error != "")
            {
                s_progress_rundata = NULL;
                throw FabberInternalError(ctx.error);
            }
        }
        s_progress_rundata = NULL;
    }
    else if (!m_store->has_device_model)
    {
        LOG << "Vb::Voxelwise calculations on the MI355X engine with the model evaluated on the host, " << m_nvoxels
            << " voxels x " << cfg.n_times << " timepoints" << endl;
        BuildInitialMvn(rundata, cfg);
        prepare_host_model();
        rc = fabber_vb_run_hostmodel_host(&cfg, series, &out, device, &Vb::LineariseCallback, &ctx);
        if (rc == -54 && ctx.error != "")
            throw FabberInternalError(ctx.error);
    }
    else
    {
        LOG << "Vb::Voxelwise calculations on the MI355X engine, kernel " << fabber_vb_kernel_name(&cfg) << ", "
            << m_nvoxels << " voxels x " << cfg.n_times << " timepoints" << endl;
        if (devices_opt != "")
        {
            fvb_summary total;
            rc = fabber_vb_run_host_multi(&cfg, series, &out, device_list.empty() ? NULL : device_list.data(),
                (int32_t)device_list.size(), &total);
            if (rc == 0)
                LOG << "Vb::devices=" << devices_opt << ": " << total.sum_iterations << " voxel-iterations, " << total.bad_voxels
                    << " voxels stopped on a numerical error" << (m_needF ? ", global free energy " : "")
                    << (m_needF ? stringify(total.sum_free_energy) : string("")) << endl;
        }
        else
            rc = fabber_vb_run_host(&cfg, series, &out, device);
    }
    if (rc != 0)
        throw FabberInternalError(string("MI355X engine failed: ") + fabber_vb_last_error());
    timer.lap("engine");

    // ---- per-voxel failures: what the reference's catch blocks do (inference_vb.cc:529-544) ----
    static const char *reasons[] = { "", "LinearizedFwdModel::ReCentre: Non-finite values found in offset",
        "LinearizedFwdModel::ReCentre: Non-finite values found in jacobian", "WhiteNoiseModel::Non-finite free energy!",
        "NEWMAT exception: matrix is singular", "Ar1cNoiseModel: negative variance" };
    int n_bad = 0;
    for (int v = 0; v < m_nvoxels; v++)
    {
        const int code = m_status[v] & 0xff;
        if (code == 0)
            continue;
        const bool in_setup = (m_status[v] & 0x100) != 0;
        const string msg = reasons[code < 6 ? code : 4];
        if (n_bad < 20)
            LOG << "Vb::Internal error for voxel " << v + 1 << " at " << coords.at0(0, v) << " " << coords.at0(1, v) << " "
                << coords.at0(2, v) << " : " << msg << endl;
        n_bad++;
        // the initial ReCentre sits outside the reference's try block (inference_vb.cc:235)
        if (m_halt_bad_voxel || in_setup)
            throw FabberInternalError(msg);
    }
    if (n_bad)
        LOG << "Vb::" << n_bad << " voxels had numerical errors and were stopped early" << endl;

    if (m_saveFsHistory)
    {
        // pad each voxel's history with its last value up to the longest (inference_vb.cc:1016-1046)
        int longest = 0;
        for (int v = 0; v < m_nvoxels; v++)
            longest = std::max(longest, std::min(hist_len[v], cfg.f_history_rows));
        Matrix hist(longest, m_nvoxels);
        for (int v = 0; v < m_nvoxels; v++)
        {
            const int n = std::min(hist_len[v], cfg.f_history_rows);
            for (int r = 0; r < longest; r++)
                hist.at0(r, v) = m_f_history.at0(r < n ? r : n - 1, v);
        }
        m_f_history = hist;
    }
    timer.lap("status pass");
}

static void save_rows(FabberRunData &rundata, const string &name, const std::vector<double, NEWMAT::DefaultInitAllocator<double> > &buf,
    int rows, int row0, int n_rows, int V)
{
    Matrix m;
    m.ReSizeNoInit(n_rows, V);
    (void)rows;
    if (n_rows > 0 && V > 0)
        memcpy(m.Store(), buf.data() + (size_t)row0 * V, sizeof(double) * (size_t)n_rows * V);
    rundata.SaveVoxelDataMove(name, m);
}

// InferenceTechnique::SaveResults of the reference (inference.cc:112-281) plus the noise images
// of Vb::SaveResults (inference_vb.cc:981-994), from the packed result image by the engine's
// post-processing kernel. N = noise entries in the MVN (0 for NLLS), n_noise_saved = how many of
// them go into noise_means / noise_stdevs.
void InferenceTechnique::SaveEngineResults(FabberRunData &rundata, const fvb_config &cfg, const vector<Parameter> &params,
    int N, int n_noise_saved, bool host_model) const
{
    FabberStageTimer timer("SaveEngineResults");
    InferenceTechnique::SaveResults(rundata); // finalMVN
    timer.lap("finalMVN");

    const int V = cfg.n_voxels, P = cfg.n_params, T = cfg.n_times;
    const bool want_mean = rundata.GetBool("save-mean"), want_std = rundata.GetBool("save-std");
    const bool want_zstat = rundata.GetBool("save-zstat"), want_var = rundata.GetBool("save-var");
    const bool want_nmean = rundata.GetBool("save-noise-mean"), want_nstd = rundata.GetBool("save-noise-std");
    const bool want_fit = rundata.GetBool("save-model-fit"), want_resid = rundata.GetBool("save-residuals");

    // (sized without being written: the post-processing fills them)
    typedef std::vector<double, NEWMAT::DefaultInitAllocator<double> > Image;
    Image mean, var, sd, zstat, fit, resid, nmean, nstd;
    fvb_postproc pp;
    memset(&pp, 0, sizeof(pp));
    const size_t PV = (size_t)P * V, TV = (size_t)T * V, NV = (size_t)N * V;
    if (want_mean)
        mean.resize(PV), pp.mean = mean.data();
    if (want_var)
        var.resize(PV), pp.var = var.data();
    if (want_std)
        sd.resize(PV), pp.std = sd.data();
    if (want_zstat)
        zstat.resize(PV), pp.zstat = zstat.data();
    if (want_fit)
        fit.resize(TV), pp.modelfit = fit.data();
    if (want_resid)
        resid.resize(TV), pp.residuals = resid.data();
    if (want_nmean && N > 0)
        nmean.resize(NV), pp.noise_mean = nmean.data();
    if (want_nstd && N > 0)
        nstd.resize(NV), pp.noise_std = nstd.data();

    if (V > 0)
    {
        if (host_model) // the model prediction can only come from the model's own host code
            pp.modelfit = pp.residuals = NULL;
        fvb_config pcfg = cfg;
        int series_rows = 0, series_cols = 0;
        // (only the residuals need the series on the device: not uploaded otherwise)
        const void *series = (pp.residuals != NULL) ? engine_series(rundata, host_model, pcfg.data_f64, series_rows, series_cols) : NULL;
        int rc = fabber_vb_postproc_host(&pcfg, series, m_result_image.Store(), &pp, rundata.GetIntDefault("device", 0, 0));
        if (rc != 0)
            throw FabberInternalError(string("MI355X engine failed in post-processing: ") + fabber_vb_last_error());
        timer.lap("post-processing kernel (host pointers)");
        if (host_model && (want_fit || want_resid)) // inference.cc:181-243
        {
            const Matrix &data = rundata.GetMainVoxelData();
            const Matrix &coords = rundata.GetVoxelCoords();
            const Matrix &supp = rundata.GetVoxelSuppData();
            const int nCov = (P + N) * (P + N + 1) / 2;
            ColumnVector tmp, means(P);
            for (int v = 0; v < V; v++)
            {
                if (supp.Ncols() == V) // as the voxel loop passes it (inference_vb.cc:104-115)
                    m_model->PassData(v + 1, ColumnVector(data.Column(v + 1)), ColumnVector(coords.Column(v + 1)), ColumnVector(supp.Column(v + 1)));
                else
                    m_model->PassData(v + 1, ColumnVector(data.Column(v + 1)), ColumnVector(coords.Column(v + 1)));
                for (int k = 0; k < P; k++)
                    means(k + 1) = m_result_image.at0(nCov + k, v);
                m_model->EvaluateFabber(means, tmp, "");
                if (tmp.Nrows() != T)
                    throw FabberInternalError("The model's prediction has " + stringify(tmp.Nrows()) + " timepoints, the data " + stringify(T));
                for (int t = 0; t < T; t++)
                {
                    if (want_fit)
                        fit[(size_t)t * V + v] = tmp(t + 1);
                    if (want_resid)
                        resid[(size_t)t * V + v] = data.at0(t, v) - tmp(t + 1);
                }
            }
        }
    }
    for (int k = 0; k < P; k++)
    {
        const string &name = params[k].name;
        if (want_mean)
            save_rows(rundata, "mean_" + name, mean, P, k, 1, V);
        if (want_zstat)
            save_rows(rundata, "zstat_" + name, zstat, P, k, 1, V);
        if (want_std)
            save_rows(rundata, "std_" + name, sd, P, k, 1, V);
        if (want_var)
            save_rows(rundata, "var_" + name, var, P, k, 1, V);
    }
    if (want_resid)
        save_rows(rundata, "residuals", resid, T, 0, T, V);
    if (want_fit)
        save_rows(rundata, "modelfit", fit, T, 0, T, V);
    if (want_nmean && N > 0)
        save_rows(rundata, "noise_means", nmean, N, 0, n_noise_saved, V); // first NumParams() noise entries (inference_vb.cc:981-989)
    if (want_nstd && N > 0)
        save_rows(rundata, "noise_stdevs", nstd, N, 0, n_noise_saved, V);
    timer.lap("images into the run data");

    // model-specific extra outputs are defined by host code only (FwdModel::EvaluateModel with
    // a key): evaluate them on the host, like inference.cc:181-252
    vector<string> outputs;
    m_model->GetOutputs(outputs);
    if (rundata.GetBool("save-model-extras") && !outputs.empty() && V > 0)
    {
        const Matrix &data = rundata.GetMainVoxelData();
        const Matrix &coords = rundata.GetVoxelCoords();
        const int nCov = (P + N) * (P + N + 1) / 2;
        for (size_t o = 0; o < outputs.size(); o++)
        {
            if (outputs[o] == "")
                continue;
            Matrix result;
            ColumnVector tmp, means(P);
            for (int v = 0; v < V; v++)
            {
                try
                {
                    m_model->PassData(v + 1, ColumnVector(data.Column(v + 1)), ColumnVector(coords.Column(v + 1)));
                    for (int k = 0; k < P; k++)
                        means(k + 1) = m_result_image.at0(nCov + k, v);
                    m_model->EvaluateFabber(means, tmp, outputs[o]);
                    if (result.Nrows() != tmp.Nrows())
                        result.ReSize(tmp.Nrows(), V);
                    result.Column(v + 1) = tmp;
                }
                catch (std::exception &e)
                {
                    LOG << "InferenceTechnique::Error generating output " << outputs[o] << " for voxel " << v + 1 << " : " << e.what() << endl;
                }
            }
            rundata.SaveVoxelData(outputs[o], result);
        }
    }

}

void Vb::SaveResults(FabberRunData &rundata) const
{
    LOG << "Vb::Preparing to save results..." << endl;
    SaveEngineResults(rundata, m_store->cfg, m_store->params, m_noise_params, m_noise->NumParams(), !m_store->has_device_model);
    const int V = m_nvoxels;
    if (m_saveF && m_needF && !m_free_energy.empty())
    {
        Matrix F(1, V);
        for (int v = 0; v < V; v++)
            F.at0(0, v) = m_free_energy[v];
        rundata.SaveVoxelData("freeEnergy", F);
    }
    else
    {
        LOG << "Vb::Free energy wasn't recorded, so no freeEnergy data saved" << endl;
    }
    if (V > 0 && m_saveFsHistory && m_f_history.Nrows() > 0)
    {
        Matrix hist = m_f_history;
        rundata.SaveVoxelData("freeEnergyHistory", hist);
    }
    LOG << "Vb::Done writing results." << endl;
}
