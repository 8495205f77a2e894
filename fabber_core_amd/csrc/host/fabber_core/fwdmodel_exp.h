/* fwdmodel_exp.h - sum of exponentials amp_i exp(-r_i t) (reference: examples/fwdmodel_exp.h).
 * Built in here (the reference ships it as an example plugin) because BASELINE configs 2, 3
 * and 5 are quoted on it and it has a device body. */
#pragma once

#include "fwdmodel.h"

class ExpFwdModel : public FwdModel
{
public:
    static FwdModel *NewInstance();
    ExpFwdModel()
        : m_num(1)
        , m_dt(1.0)
    {
    }
    std::string ModelVersion() const;
    std::string GetDescription() const;
    void GetOptions(std::vector<OptionSpec> &opts) const;
    void Initialize(FabberRunData &args);
    void EvaluateModel(
        const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result, const std::string &key = "") const;
    void InitVoxelPosterior(MVNDist &posterior) const;
    bool GetDeviceModel(DeviceModelSpec &spec) const;

protected:
    void GetParameterDefaults(std::vector<Parameter> &params) const;

private:
    int m_num;
    double m_dt;
};
