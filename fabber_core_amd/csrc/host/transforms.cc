// transforms.cc - host twins of the device transform functions; the arithmetic is shared with
// the kernels through vb_math.h so that host-resolved priors and device-evaluated models can
// never drift apart.
#include "transforms.h"

#include "../vb_math.h"

double Transform::ToModelVar(double val) const
{
    return fvb::to_model_var(DeviceCode(), val);
}
double Transform::ToFabberVar(double val) const
{
    return fvb::to_fabber_var(DeviceCode(), val);
}
DistParams Transform::ToModel(DistParams p) const
{
    return DistParams(ToModel(p.mean()), ToModelVar(p.var()));
}
DistParams Transform::ToFabber(DistParams p) const
{
    return DistParams(ToFabber(p.mean()), ToFabberVar(p.var()));
}

#define FVB_TRANSFORM_IMPL(CLASS, CODE)                                                                      \
    int CLASS::DeviceCode() const                                                                            \
    {                                                                                                        \
        return CODE;                                                                                         \
    }                                                                                                        \
    double CLASS::ToModel(double val) const                                                                  \
    {                                                                                                        \
        return fvb::to_model(CODE, val);                                                                     \
    }                                                                                                        \
    double CLASS::ToFabber(double val) const                                                                 \
    {                                                                                                        \
        return fvb::to_fabber(CODE, val);                                                                    \
    }
#define FVB_TRANSFORM_VAR_IMPL(CLASS, CODE)                                                                  \
    double CLASS::ToModelVar(double val) const                                                               \
    {                                                                                                        \
        return fvb::to_model_var(CODE, val);                                                                 \
    }                                                                                                        \
    double CLASS::ToFabberVar(double val) const                                                              \
    {                                                                                                        \
        return fvb::to_fabber_var(CODE, val);                                                                \
    }

FVB_TRANSFORM_IMPL(IdentityTransform, FVB_TRANSFORM_IDENTITY)
FVB_TRANSFORM_VAR_IMPL(IdentityTransform, FVB_TRANSFORM_IDENTITY)
FVB_TRANSFORM_IMPL(LogTransform, FVB_TRANSFORM_LOG)
FVB_TRANSFORM_VAR_IMPL(LogTransform, FVB_TRANSFORM_LOG)
FVB_TRANSFORM_IMPL(SoftPlusTransform, FVB_TRANSFORM_SOFTPLUS)
FVB_TRANSFORM_IMPL(FractionalTransform, FVB_TRANSFORM_FRACTIONAL)
FVB_TRANSFORM_VAR_IMPL(FractionalTransform, FVB_TRANSFORM_FRACTIONAL)
FVB_TRANSFORM_IMPL(AbsTransform, FVB_TRANSFORM_ABS)

const Transform *TRANSFORM_IDENTITY()
{
    static IdentityTransform t;
    return &t;
}
const Transform *TRANSFORM_LOG()
{
    static LogTransform t;
    return &t;
}
const Transform *TRANSFORM_SOFTPLUS()
{
    static SoftPlusTransform t;
    return &t;
}
const Transform *TRANSFORM_FRACTIONAL()
{
    static FractionalTransform t;
    return &t;
}
const Transform *TRANSFORM_ABS()
{
    static AbsTransform t;
    return &t;
}

const Transform *GetTransform(std::string id)
{
    if (id == TRANSFORM_CODE_IDENTITY)
        return TRANSFORM_IDENTITY();
    if (id == TRANSFORM_CODE_LOG)
        return TRANSFORM_LOG();
    if (id == TRANSFORM_CODE_SOFTPLUS)
        return TRANSFORM_SOFTPLUS();
    if (id == TRANSFORM_CODE_FRACTIONAL)
        return TRANSFORM_FRACTIONAL();
    if (id == TRANSFORM_CODE_ABS)
        return TRANSFORM_ABS();
    throw InvalidOptionValue("PSP_byname<n>_transform", id, "Supported transforms: I, L, S, F, A");
}
