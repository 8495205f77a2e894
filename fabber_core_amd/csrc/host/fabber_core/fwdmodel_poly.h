/* fwdmodel_poly.h - polynomial test model c0 + c1 x + ... (reference: fwdmodel_poly.h) */
#pragma once

#include "fwdmodel.h"

class PolynomialFwdModel : public FwdModel
{
public:
    static FwdModel *NewInstance();
    void GetOptions(std::vector<OptionSpec> &opts) const;
    std::string GetDescription() const;
    std::string ModelVersion() const;
    void Initialize(FabberRunData &args);
    void EvaluateModel(
        const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result, const std::string &key = "") const;
    bool GetDeviceModel(DeviceModelSpec &spec) const;

protected:
    void GetParameterDefaults(std::vector<Parameter> &params) const;

private:
    int m_degree;
};
