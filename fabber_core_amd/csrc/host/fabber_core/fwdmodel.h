/* fwdmodel.h - forward-model plugin interface.
 *
 * Same virtual surface, protected members and registration idiom as the reference's FwdModel
 * (fwdmodel.h:24-383), so that existing model sources (e.g. examples/fwdmodel_exp.cc) compile
 * unchanged against this header. One addition: GetDeviceModel(), through which a model tells
 * the MI355X engine that it has a device body (the built-in poly / linear / exp models do); a
 * model that does not override it is still usable - its Jacobian is then produced on the host by
 * calling EvaluateModel (slow path, see inference_vb.cc). */
#pragma once

#include "dist_mvn.h"
#include "easylog.h"
#include "factories.h"
#include "transforms.h"

#include "armawrap/newmat.h"

#include <map>
#include <string>
#include <vector>

/** Everything fabber knows about one model parameter */
struct Parameter
{
    Parameter(unsigned int idx, std::string name = "", DistParams prior = DistParams(0, 1),
        DistParams post = DistParams(0, 1), char prior_type = 'N', const Transform *transform = TRANSFORM_IDENTITY(),
        std::string desc = "", std::string units = "")
        : idx(idx)
        , name(name)
        , desc(desc)
        , units(units)
        , prior(prior)
        , post(post)
        , prior_type(prior_type)
        , transform(transform)
    {
    }

    unsigned int idx;
    std::string name, desc, units;
    DistParams prior;
    DistParams post;
    char prior_type;
    const Transform *transform;
    /** run-time extras resolved from options, e.g. "image" -> data key of an image prior */
    std::map<std::string, std::string> options;
};

/** Description of a model's device body for the HIP engine (enum fvb_model + options). */
struct DeviceModelSpec
{
    DeviceModelSpec()
        : model(-1)
    {
        for (int i = 0; i < 4; i++)
        {
            iopt[i] = 0;
            dopt[i] = 0;
        }
    }
    int model;
    int iopt[4];
    double dopt[4];
    NEWMAT::Matrix design; // linear model: T x P
};

class FwdModel : public Loggable
{
public:
    virtual ~FwdModel()
    {
    }
    virtual std::string GetDescription() const;
    virtual std::string ModelVersion() const;
    virtual void GetOptions(std::vector<OptionSpec> &opts) const
    {
    }
    virtual void Initialize(FabberRunData &rundata);
    virtual void GetOutputs(std::vector<std::string> &outputs) const
    {
    }
    /** Per-voxel initial posterior in MODEL space (transforms are applied afterwards) */
    virtual void InitVoxelPosterior(MVNDist &posterior) const
    {
        InitParams(posterior);
    }
    /** Model prediction for MODEL-space parameters */
    virtual void EvaluateModel(
        const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result, const std::string &key = "") const
    {
        Evaluate(params, result);
    }

    /** MI355X extension: describe the device body of this model. Default: none. */
    virtual bool GetDeviceModel(DeviceModelSpec &spec) const
    {
        return false;
    }

    void GetParameters(FabberRunData &rundata, std::vector<Parameter> &params);
    void PassData(unsigned int voxel_idx, const NEWMAT::ColumnVector &voxdata, const NEWMAT::ColumnVector &coords,
        const NEWMAT::ColumnVector &voxsuppdata = NEWMAT::ColumnVector());
    void GetInitialPosterior(MVNDist &posterior, FabberRunData &rundata) const;
    /** Model prediction for FABBER-space parameters (applies the transforms) */
    void EvaluateFabber(
        const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result, const std::string &key = "") const;
    void ToFabber(MVNDist &mvn) const;
    void ToModel(MVNDist &mvn) const;

    static void LoadFromDynamicLibrary(const std::string &filename, EasyLog *log = 0);
    static std::vector<std::string> GetKnown();
    static FwdModel *NewFromName(const std::string &name);
    static void UsageFromName(const std::string &name, std::ostream &stream);

#ifdef DEPRECATED
    virtual void InitParams(MVNDist &posterior) const
    {
    }
    virtual void Evaluate(const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result) const
    {
    }
    virtual int NumParams() const
    {
        return m_params.size();
    }
    virtual void NameParams(std::vector<std::string> &names) const
    {
    }
    virtual void HardcodedInitialDists(MVNDist &prior, MVNDist &posterior) const {};
    virtual void UpdateARD(const MVNDist &posterior, MVNDist &prior, double &Fard) const
    {
    }
    virtual void SetupARD(const MVNDist &posterior, MVNDist &prior, double &Fard) const
    {
    }
    std::vector<int> ardindices;
    virtual void DumpParameters(const NEWMAT::ColumnVector &params, const std::string &indent = "") const;
    virtual void Usage(std::ostream &stream) const;
#endif

protected:
    virtual void GetParameterDefaults(std::vector<Parameter> &params) const;

    // current voxel (PassData)
    unsigned int voxel;
    NEWMAT::ColumnVector coords;
    NEWMAT::ColumnVector data;
    NEWMAT::ColumnVector suppdata;
#ifdef DEPRECATED
    int coord_x;
    int coord_y;
    int coord_z;
#endif
    std::vector<Parameter> m_params;
};

typedef SingletonFactory<FwdModel> FwdModelFactory;

/** Signature of the constructor functions exported by model libraries */
typedef FwdModel *(*NewInstanceFptr)(void);
