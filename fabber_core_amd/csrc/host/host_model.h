// host_model.h - forward models that exist only as host code (a FwdModel of a model library written for the
// reference), shared by the techniques that hand their evaluations to the engine as a callback
// (fvb_linearise_fn of include/fabber_vb.h): method=vb / spatialvb (inference_vb.cc) and method=nlls
// (inference_nlls.cc). Internal to the host library.
#pragma once

#include "fwdmodel.h"
#include "rundata.h"

#include <memory>
#include <string>
#include <vector>

struct HostModelContext
{
    void *self;
    FwdModel *model;
    FabberRunData *rundata;
    const NEWMAT::Matrix *data, *coords, *suppdata;
    int T, P;
    std::string error;
    std::vector<FwdModel *> models; // [0] = the technique's own instance, the rest are per-thread copies
};

/** One model instance per host thread (option host-model-threads); fills ctx.models and ctx.data. Returns the count. */
int host_model_instances(HostModelContext &ctx, std::vector<std::unique_ptr<FwdModel> > &copies, FwdModel *model, FabberRunData &rundata,
    EasyLog *log);

/** fvb_linearise_fn: LinearizedFwdModel::ReCentre (fwdmodel_linear.cc:126-182) for the active voxels, shared out over
 *  the instances; user = HostModelContext*. An exception of the model ends up in ctx.error, return value 1. */
int32_t host_model_linearise(void *user, int32_t n_active, const int32_t *ids, const double *means, double *lin);
