/* fwdmodel_linear.h - design-matrix model ("linear") and the linearisation value type.
 * Surface of the reference's LinearFwdModel / LinearizedFwdModel (fwdmodel_linear.h:23-144).
 * In this build LinearizedFwdModel::ReCentre is only used by host-side tools and by the
 * host-Jacobian path for models without a device body; the built-in models are re-linearised
 * inside the HIP kernels. */
#pragma once

#include "fwdmodel.h"

#include "armawrap/newmat.h"

#include <string>
#include <vector>

class LinearFwdModel : public FwdModel
{
public:
    static FwdModel *NewInstance();
    LinearFwdModel()
    {
    }
    virtual ~LinearFwdModel()
    {
    }
    virtual void GetOptions(std::vector<OptionSpec> &opts) const;
    virtual std::string GetDescription() const;
    virtual std::string ModelVersion() const;
    virtual void Initialize(FabberRunData &args);
    virtual void EvaluateModel(
        const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result, const std::string &key = "") const;
    virtual bool GetDeviceModel(DeviceModelSpec &spec) const;

    NEWMAT::ReturnMatrix Jacobian() const;
    NEWMAT::ReturnMatrix Centre() const;
    NEWMAT::ReturnMatrix Offset() const;

protected:
    virtual void GetParameterDefaults(std::vector<Parameter> &params) const;
    NEWMAT::Matrix m_jacobian;   // J (tranposed?) : T x P
    NEWMAT::ColumnVector m_centre; // m
    NEWMAT::ColumnVector m_offset; // g(m)
};

/** Linear approximation g(m) + J (theta - m) of another model about a movable centre */
class LinearizedFwdModel : public LinearFwdModel
{
public:
    explicit LinearizedFwdModel(const FwdModel *model);
    LinearizedFwdModel(const LinearizedFwdModel &from);
    /** central differences, step 1e-5 |m_i| with floor 1e-10; throws FabberInternalError on
     *  non-finite prediction / Jacobian */
    void ReCentre(const NEWMAT::ColumnVector &about);
    virtual void GetParameterDefaults(std::vector<Parameter> &params) const
    {
        assert(false);
    }

private:
    const FwdModel *m_model;
};
