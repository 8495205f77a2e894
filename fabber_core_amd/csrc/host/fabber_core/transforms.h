/* transforms.h - parameter transforms between model space and fabber's Gaussian space.
 * Host-side twins of the device functions in ../../vb_math.h; public surface as the reference's
 * transforms.h:19-259 (DistParams, Transform hierarchy, TRANSFORM_* singletons, GetTransform). */
#pragma once

#include "rundata.h"

#include <math.h>
#include <string>

const std::string TRANSFORM_CODE_IDENTITY = "I";
const std::string TRANSFORM_CODE_LOG = "L";
const std::string TRANSFORM_CODE_SOFTPLUS = "S";
const std::string TRANSFORM_CODE_FRACTIONAL = "F";
const std::string TRANSFORM_CODE_ABS = "A";

/** Mean / variance / precision of one parameter's Gaussian */
struct DistParams
{
    DistParams(double m = 0, double v = 1)
        : m_mean(m)
        , m_var(v)
        , m_prec(1 / v)
    {
    }
    double mean() const
    {
        return m_mean;
    }
    double var() const
    {
        return m_var;
    }
    double prec() const
    {
        return m_prec;
    }

private:
    double m_mean, m_var, m_prec;
};

class Transform
{
public:
    virtual ~Transform()
    {
    }
    /** engine code: one of enum fvb_transform (include/fabber_vb.h) */
    virtual int DeviceCode() const = 0;
    virtual double ToModel(double val) const = 0;
    virtual double ToFabber(double val) const = 0;
    virtual double ToModelVar(double val) const;
    virtual double ToFabberVar(double val) const;
    DistParams ToModel(DistParams params) const;
    DistParams ToFabber(DistParams params) const;
};

class IdentityTransform : public Transform
{
public:
    int DeviceCode() const;
    double ToModel(double val) const;
    double ToFabber(double val) const;
    double ToModelVar(double val) const;
    double ToFabberVar(double val) const;
};
class LogTransform : public Transform
{
public:
    int DeviceCode() const;
    double ToModel(double val) const;
    double ToFabber(double val) const;
    double ToModelVar(double val) const;
    double ToFabberVar(double val) const;
};
class SoftPlusTransform : public Transform
{
public:
    int DeviceCode() const;
    double ToModel(double val) const;
    double ToFabber(double val) const;
};
class FractionalTransform : public Transform
{
public:
    int DeviceCode() const;
    double ToModel(double val) const;
    double ToFabber(double val) const;
    double ToModelVar(double val) const;
    double ToFabberVar(double val) const;
};
class AbsTransform : public Transform
{
public:
    int DeviceCode() const;
    double ToModel(double val) const;
    double ToFabber(double val) const;
};

const Transform *TRANSFORM_IDENTITY();
const Transform *TRANSFORM_LOG();
const Transform *TRANSFORM_SOFTPLUS();
const Transform *TRANSFORM_FRACTIONAL();
const Transform *TRANSFORM_ABS();
const Transform *GetTransform(std::string id);
