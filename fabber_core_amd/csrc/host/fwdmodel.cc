// fwdmodel.cc - forward-model base class: parameter resolution, transforms, plugin loading.
// Behaviour follows the reference's fwdmodel.cc (GetParameters :210-282, GetInitialPosterior
// :284-313, ToFabber/ToModel :315-337, EvaluateFabber :365-382, LoadFromDynamicLibrary :63-129).
#include "fwdmodel.h"

#include "priors.h"

#include <algorithm>
#include <dlfcn.h>
#include <memory>

using namespace std;
using NEWMAT::ColumnVector;
using NEWMAT::SymmetricMatrix;

typedef int (*GetNumModelsFptr)(void);
typedef const char *(*GetModelNameFptr)(int);
typedef NewInstanceFptr (*GetNewInstanceFptrFptr)(const char *);

void FwdModel::LoadFromDynamicLibrary(const std::string &filename, EasyLog *log)
{
    if (log)
        log->LogStream() << "FwdModel::Loading dynamic models from " << filename << endl;
    void *lib = dlopen(filename.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!lib)
        throw InvalidOptionValue("loadmodels", filename, string("Failed to open library ") + dlerror());
    GetNumModelsFptr get_num_models = (GetNumModelsFptr)dlsym(lib, "get_num_models");
    GetModelNameFptr get_model_name = (GetModelNameFptr)dlsym(lib, "get_model_name");
    GetNewInstanceFptrFptr get_new_instance_fptr = (GetNewInstanceFptrFptr)dlsym(lib, "get_new_instance_func");
    if (!get_num_models || !get_model_name || !get_new_instance_fptr)
        throw InvalidOptionValue("loadmodels", filename,
            "Library does not export get_num_models / get_model_name / get_new_instance_func");
    const int n = get_num_models();
    FwdModelFactory *factory = FwdModelFactory::GetInstance();
    for (int i = 0; i < n; i++)
    {
        const char *name = get_model_name(i);
        if (!name)
            throw InvalidOptionValue("loadmodels", filename, "Dynamic library failed to return model name for index " + stringify(i));
        NewInstanceFptr maker = get_new_instance_fptr(name);
        if (!maker)
            throw InvalidOptionValue("loadmodels", filename, string("Dynamic library failed to return new instance function for model ") + name);
        if (log)
            log->LogStream() << "FwdModel::Loading model " << name << endl;
        factory->Add(name, maker);
    }
}

std::vector<std::string> FwdModel::GetKnown()
{
    return FwdModelFactory::GetInstance()->GetNames();
}

FwdModel *FwdModel::NewFromName(const string &name)
{
    FwdModel *model = FwdModelFactory::GetInstance()->Create(name);
    if (!model)
        throw InvalidOptionValue("model", name, "Unrecognized forward model");
    return model;
}

void FwdModel::Initialize(FabberRunData &args)
{
    m_log = args.GetLogger();
}

void FwdModel::UsageFromName(const string &name, std::ostream &stream)
{
    std::unique_ptr<FwdModel> model(NewFromName(name));
    stream << name << ": " << model->ModelVersion() << endl << endl << model->GetDescription() << endl << endl;
    stream << "Options: " << endl << endl;
    vector<OptionSpec> options;
    model->GetOptions(options);
    if (options.empty())
        model->Usage(stream);
    for (size_t i = 0; i < options.size(); i++)
        stream << options[i];
    vector<string> outputs;
    model->GetOutputs(outputs);
    if (!outputs.empty())
    {
        stream << endl << "Additional outputs: " << endl << endl;
        for (size_t i = 0; i < outputs.size(); i++)
            if (outputs[i] != "")
                stream << "  " << outputs[i] << endl;
    }
}

string FwdModel::GetDescription() const
{
    return "No description available";
}
string FwdModel::ModelVersion() const
{
    return "No version info available.";
}
void FwdModel::Usage(std::ostream &stream) const
{
    stream << "No usage information available" << endl;
}

void FwdModel::PassData(unsigned int voxel_idx, const ColumnVector &voxdata, const ColumnVector &voxcoords,
    const ColumnVector &voxsuppdata)
{
    voxel = voxel_idx;
    data = voxdata;
    suppdata = voxsuppdata;
    coords = voxcoords;
    coord_x = (int)coords(1);
    coord_y = (int)coords(2);
    coord_z = (int)coords(3);
}

void FwdModel::GetParameters(FabberRunData &rundata, vector<Parameter> &params)
{
    GetParameterDefaults(params);
    m_params.clear();
    const string types
        = Prior::ExpandPriorTypesString(rundata.GetStringDefault("param-spatial-priors", ""), params.size());

    for (size_t k = 0; k < params.size(); k++)
    {
        Parameter &p = params[k];
        // (1) positional prior types from param-spatial-priors
        if (types[p.idx] != PRIOR_DEFAULT)
            p.prior_type = types[p.idx];
        p.options["image"] = "image-prior" + stringify(p.idx + 1);

        // (2) PSP_byname<n> blocks addressed to this parameter override (1)
        for (int n = 1;; n++)
        {
            const string stem = "PSP_byname" + stringify(n);
            const string name = rundata.GetStringDefault(stem, "stop!");
            if (name == "stop!")
                break;
            if (name != p.name)
                continue;
            const string tcode = rundata.GetStringDefault(stem + "_transform", "");
            if (tcode != "")
                p.transform = GetTransform(tcode);
            const char ptype = convertTo<char>(rundata.GetStringDefault(stem + "_type", stringify(p.prior_type)));
            if (ptype != PRIOR_DEFAULT)
                p.prior_type = ptype;
            const double mean = rundata.GetDoubleDefault(stem + "_mean", p.prior.mean());
            const double prec = rundata.GetDoubleDefault(stem + "_prec", p.prior.prec());
            p.prior = DistParams(mean, 1 / prec);
            p.options["image"] = stem + "_image";
        }

        // (3) cap the precision, (4) move the prior into fabber space
        if (p.prior.prec() > 1e12)
        {
            WARN_ONCE("Specified precision " + stringify(p.prior.prec())
                + " is very high - this can trigger numerical instability. Using 1e12 instead");
            p.prior = DistParams(p.prior.mean(), 1e-12);
        }
        p.prior = p.transform->ToFabber(p.prior);
        m_params.push_back(p);
    }
}

void FwdModel::GetInitialPosterior(MVNDist &posterior, FabberRunData &rundata) const
{
    posterior.SetSize(m_params.size());
    SymmetricMatrix cov = posterior.GetCovariance();
    for (size_t p = 0; p < m_params.size(); p++)
    {
        if (m_params[p].prior_type == PRIOR_IMAGE)
        {
            const string key = m_params[p].options.find("image")->second;
            posterior.means(p + 1) = rundata.GetVoxelData(key)(1, voxel);
        }
        else
        {
            posterior.means(p + 1) = m_params[p].post.mean();
        }
        cov(p + 1, p + 1) = m_params[p].post.var();
    }
    posterior.SetCovariance(cov);
    InitVoxelPosterior(posterior);
    ToFabber(posterior);
}

void FwdModel::ToFabber(MVNDist &mvn) const
{
    SymmetricMatrix cov = mvn.GetCovariance();
    for (size_t p = 0; p < m_params.size(); p++)
    {
        mvn.means(p + 1) = m_params[p].transform->ToFabber(mvn.means(p + 1));
        cov(p + 1, p + 1) = m_params[p].transform->ToFabberVar(cov(p + 1, p + 1));
    }
    mvn.SetCovariance(cov);
}

void FwdModel::ToModel(MVNDist &mvn) const
{
    SymmetricMatrix cov = mvn.GetCovariance();
    for (size_t p = 0; p < m_params.size(); p++)
    {
        DistParams dp = m_params[p].transform->ToModel(DistParams(mvn.means(p + 1), cov(p + 1, p + 1)));
        mvn.means(p + 1) = dp.mean();
        cov(p + 1, p + 1) = dp.var();
    }
    mvn.SetCovariance(cov);
}

// Fallback for models written against the pre-Parameter API (NameParams + HardcodedInitialDists
// + ardindices), fwdmodel.cc:339-363
void FwdModel::GetParameterDefaults(vector<Parameter> &params) const
{
    params.clear();
    vector<string> names;
    NameParams(names);
    if (names.empty())
        return;
    MVNDist priors(names.size()), posts(names.size());
    HardcodedInitialDists(priors, posts);
    for (unsigned int i = 0; i < names.size(); i++)
    {
        DistParams prior(priors.means(i + 1), priors.GetCovariance()(i + 1, i + 1));
        DistParams post(posts.means(i + 1), posts.GetCovariance()(i + 1, i + 1));
        Parameter p(i, names[i], prior, post, PRIOR_NORMAL, TRANSFORM_IDENTITY());
        if (std::find(ardindices.begin(), ardindices.end(), (int)i + 1) != ardindices.end())
            p.prior_type = PRIOR_ARD;
        params.push_back(p);
    }
}

void FwdModel::EvaluateFabber(const ColumnVector &params, ColumnVector &result, const std::string &key) const
{
    if (m_params.empty())
    {
        EvaluateModel(params, result, key);
        return;
    }
    ColumnVector tparams(params.Nrows());
    for (int i = 1; i <= params.Nrows(); i++)
        tparams(i) = m_params[i - 1].transform->ToModel(params(i));
    EvaluateModel(tparams, result, key);
}

void FwdModel::DumpParameters(const ColumnVector &params, const string &indent) const
{
    vector<string> names;
    NameParams(names);
    LOG << indent << "Parameters:" << endl;
    for (size_t i = 1; i <= names.size() && (int)i <= params.Nrows(); i++)
        LOG << indent << "  " << names[i - 1] << " = " << params(i) << endl;
    LOG << indent << "Total of " << names.size() << " parameters" << endl;
}
