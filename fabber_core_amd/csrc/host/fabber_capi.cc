// fabber_capi.cc - the C ABI (include/fabber_capi.h) over FabberRunDataArray.
// Error convention and edge cases follow the reference's fabber_capi.cc: every C++ exception is
// caught at the boundary; DataNotFound maps to -1; NEWMAT exceptions from a run map to
// FABBER_ERR_NEWMAT; a NULL mask is rejected (fabber_capi.cc:83-84); dorun needs both buffers.
#include "../../../include/fabber_capi.h"

#include "easylog.h"
#include "fwdmodel.h"
#include "inference.h"
#include "rundata_array.h"
#include "setup.h"

#include "../../../include/fabber_vb.h"

#include <algorithm>
#include <atomic>
#include <memory>
#include <sstream>
#include <string.h>

using namespace std;

namespace
{
int report(int code, const char *msg, char *err_buf)
{
    if (err_buf)
    {
        strncpy(err_buf, msg ? msg : "NULL message", FABBER_ERR_MAXC - 1);
        err_buf[FABBER_ERR_MAXC - 1] = '\0';
    }
    return code;
}

// Copy text into a caller buffer; "Buffer too small" (-1) if it does not fit with its NUL
int deliver(const string &text, unsigned int bufsize, char *buf, char *err_buf)
{
    if (text.size() >= bufsize)
        return report(-1, "Buffer too small", err_buf);
    memcpy(buf, text.c_str(), text.size() + 1);
    return 0;
}

FabberRunDataArray *ctx(void *fab)
{
    return static_cast<FabberRunDataArray *>(fab);
}

// Run `body`, translating exceptions; `what` names the operation for unknown exceptions
template <class F> int guarded(char *err_buf, const char *what, F body)
{
    try
    {
        return body();
    }
    catch (DataNotFound &e)
    {
        return report(-1, e.what(), err_buf);
    }
    catch (NEWMAT::Exception &e)
    {
        return report(FABBER_ERR_FATAL, e.what(), err_buf);
    }
    catch (std::exception &e)
    {
        return report(FABBER_ERR_FATAL, e.what(), err_buf);
    }
    catch (...)
    {
        return report(FABBER_ERR_FATAL, what, err_buf);
    }
}

// A configured instance of the model named by the "model" option
FwdModel *configured_model(FabberRunDataArray *rundata, EasyLog *log)
{
    std::unique_ptr<FwdModel> model(FwdModel::NewFromName(rundata->GetString("model")));
    model->SetLogger(log);
    model->Initialize(*rundata);
    return model.release();
}
}

static std::atomic<int> g_live_handles(0);

extern "C" {

void *fabber_new(char *err_buf)
{
    try
    {
        FabberSetup::SetupDefaults();
        void *fab = new FabberRunDataArray(false);
        ++g_live_handles;
        return fab;
    }
    catch (...)
    {
        report(FABBER_ERR_FATAL, "Failed to allocate memory for run data", err_buf);
        return NULL;
    }
}

int fabber_load_models(void *fab, const char *libpath, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!libpath)
        return report(FABBER_ERR_FATAL, "Library path is NULL", err_buf);
    return guarded(err_buf, "Error loading models", [&]() {
        FwdModel::LoadFromDynamicLibrary(libpath);
        return 0;
    });
}

int fabber_set_extent(void *fab, unsigned int nx, unsigned int ny, unsigned int nz, const int *mask, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!mask)
        return report(FABBER_ERR_FATAL, "Mask is NULL", err_buf);
    if (nx == 0 || ny == 0 || nz == 0)
        return report(FABBER_ERR_FATAL, "Dimensions must be >0", err_buf);
    return guarded(err_buf, "Error setting extent", [&]() {
        ctx(fab)->SetExtent(nx, ny, nz, mask);
        return 0;
    });
}

void fabber_destroy(void *fab)
{
    if (fab)
    {
        FabberSetup::Destroy();
        delete ctx(fab);
        // the engine keeps its device work buffers between runs (a private memory pool per device); when the last
        // handle of the process goes, all but 4 GiB per device go back to the driver (FVB_POOL_KEEP_BYTES)
        if (--g_live_handles == 0)
        {
            const char *keep = getenv("FVB_POOL_KEEP_BYTES");
            fabber_vb_trim_cached_memory(keep ? strtoull(keep, NULL, 10) : (4ull << 30));
        }
    }
}

void fabber_amd_trim_host_cache(unsigned long long keep_bytes)
{
    NEWMAT::BigBlockCache::instance().trim((size_t)keep_bytes);
}

int fabber_set_opt(void *fab, const char *key, const char *value, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!key || !value)
        return report(FABBER_ERR_FATAL, "Option key or value is NULL", err_buf);
    return guarded(err_buf, "Error setting option", [&]() {
        ctx(fab)->Set(key, string(value));
        return 0;
    });
}

int fabber_set_data(void *fab, const char *name, unsigned int data_size, const float *data, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!data)
        return report(FABBER_ERR_FATAL, "Data buffer is NULL", err_buf);
    if (!name)
        return report(FABBER_ERR_FATAL, "Data name is NULL", err_buf);
    if (data_size == 0)
        return report(FABBER_ERR_FATAL, "Data size must be >0", err_buf);
    EasyLog quiet;
    ctx(fab)->SetLogger(&quiet);
    int rc = guarded(err_buf, "Error setting data", [&]() {
        ctx(fab)->SetVoxelDataArray(name, (int)data_size, data);
        return 0;
    });
    ctx(fab)->SetLogger(NULL);
    return rc == -1 ? FABBER_ERR_FATAL : rc;
}

int fabber_get_data_size(void *fab, const char *name, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!name)
        return report(FABBER_ERR_FATAL, "Data name is NULL", err_buf);
    EasyLog quiet;
    ctx(fab)->SetLogger(&quiet);
    int rc = guarded(err_buf, "Error getting data", [&]() { return ctx(fab)->GetVoxelDataSize(name); });
    ctx(fab)->SetLogger(NULL);
    return rc;
}

int fabber_get_data(void *fab, const char *name, float *data_buf, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!name)
        return report(FABBER_ERR_FATAL, "Data name is NULL", err_buf);
    if (!data_buf)
        return report(FABBER_ERR_FATAL, "Data buffer is NULL", err_buf);
    EasyLog quiet;
    ctx(fab)->SetLogger(&quiet);
    int rc = guarded(err_buf, "Error getting data", [&]() {
        ctx(fab)->GetVoxelDataArray(name, data_buf);
        return 0;
    });
    ctx(fab)->SetLogger(NULL);
    return rc;
}

int fabber_dorun(void *fab, unsigned int log_bufsize, char *log_buf, char *err_buf, void (*progress_cb)(int, int))
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!log_buf)
        return report(FABBER_ERR_FATAL, "Log buffer is NULL", err_buf);
    if (!err_buf)
        return report(FABBER_ERR_FATAL, "Error buffer is NULL", err_buf);

    EasyLog log;
    stringstream logstr;
    FabberRunDataArray *rundata = ctx(fab);
    rundata->SetLogger(&log);
    int ret = 0;
    try
    {
        log.StartLog(logstr);
        if (progress_cb)
        {
            CallbackProgressCheck prog(progress_cb);
            rundata->Run(&prog);
        }
        else
        {
            rundata->Run();
        }
        log.ReissueWarnings();
    }
    catch (const FabberError &e)
    {
        log.ReissueWarnings();
        log.LogStream() << e.what() << endl;
        ret = report(FABBER_ERR_FATAL, e.what(), err_buf);
    }
    catch (NEWMAT::Exception &e)
    {
        log.ReissueWarnings();
        log.LogStream() << "NEWMAT exception caught in fabber:\n  " << e.what() << endl;
        ret = report(FABBER_ERR_NEWMAT, e.what(), err_buf);
    }
    catch (const std::exception &e)
    {
        log.ReissueWarnings();
        log.LogStream() << "STL exception caught in fabber:\n  " << e.what() << endl;
        ret = report(FABBER_ERR_FATAL, e.what(), err_buf);
    }
    catch (...)
    {
        log.ReissueWarnings();
        log.LogStream() << "Some other exception caught in fabber!" << endl;
        ret = report(FABBER_ERR_FATAL, "Unrecognized exception", err_buf);
    }
    log.StopLog();
    rundata->SetLogger(NULL);
    if (log_bufsize > 0)
    {
        strncpy(log_buf, logstr.str().c_str(), log_bufsize - 1);
        log_buf[log_bufsize - 1] = '\0';
    }
    return ret;
}

int fabber_get_options(void *fab, const char *key, const char *value, unsigned int out_bufsize, char *out_buf, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!out_buf)
        return report(FABBER_ERR_FATAL, "Output buffer is NULL", err_buf);
    if (key && !value)
        return report(FABBER_ERR_FATAL, "Key specified but no value", err_buf);
    return guarded(err_buf, "Error in get_options", [&]() {
        vector<OptionSpec> options;
        string desc;
        if (!key || strlen(key) == 0)
        {
            FabberRunData::GetOptions(options);
        }
        else if (strcmp(key, "model") == 0)
        {
            std::unique_ptr<FwdModel> model(FwdModel::NewFromName(value));
            desc = model->GetDescription();
            model->GetOptions(options);
        }
        else if (strcmp(key, "method") == 0)
        {
            std::unique_ptr<InferenceTechnique> method(InferenceTechnique::NewFromName(value));
            desc = method->GetDescription();
            method->GetOptions(options);
        }
        desc.erase(std::remove(desc.begin(), desc.end(), '\n'), desc.end());
        stringstream out;
        out << desc << endl;
        for (size_t i = 0; i < options.size(); i++)
            out << options[i].name << "\t" << options[i].description << "\t" << options[i].type << "\t"
                << options[i].optional << "\t" << options[i].def << endl;
        return deliver(out.str(), out_bufsize, out_buf, err_buf);
    });
}

static int deliver_list(const vector<string> &items, unsigned int out_bufsize, char *out_buf, char *err_buf)
{
    string text;
    for (size_t i = 0; i < items.size(); i++)
        text += items[i] + "\n";
    return deliver(text, out_bufsize, out_buf, err_buf);
}

int fabber_get_models(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!out_buf)
        return report(FABBER_ERR_FATAL, "Output buffer is NULL", err_buf);
    return guarded(err_buf, "Error in get_models",
        [&]() { return deliver_list(FwdModel::GetKnown(), out_bufsize, out_buf, err_buf); });
}

int fabber_get_methods(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!out_buf)
        return report(FABBER_ERR_FATAL, "Output buffer is NULL", err_buf);
    return guarded(err_buf, "Error in get_methods",
        [&]() { return deliver_list(InferenceTechnique::GetKnown(), out_bufsize, out_buf, err_buf); });
}

int fabber_get_model_params(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!out_buf)
        return report(FABBER_ERR_FATAL, "Output buffer is NULL", err_buf);
    return guarded(err_buf, "Error in get_model_params", [&]() {
        EasyLog quiet;
        std::unique_ptr<FwdModel> model(configured_model(ctx(fab), &quiet));
        vector<Parameter> params;
        model->GetParameters(*ctx(fab), params);
        vector<string> names;
        for (size_t i = 0; i < params.size(); i++)
            names.push_back(params[i].name);
        return deliver_list(names, out_bufsize, out_buf, err_buf);
    });
}

int fabber_get_model_param_descs(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!out_buf)
        return report(FABBER_ERR_FATAL, "Output buffer is NULL", err_buf);
    return guarded(err_buf, "Error in get_model_params", [&]() {
        EasyLog quiet;
        std::unique_ptr<FwdModel> model(configured_model(ctx(fab), &quiet));
        vector<Parameter> params;
        model->GetParameters(*ctx(fab), params);
        vector<string> lines;
        for (size_t i = 0; i < params.size(); i++)
        {
            string line = params[i].name + " " + params[i].desc;
            if (params[i].units != "")
                line += " (units: " + params[i].units + ")";
            lines.push_back(line);
        }
        return deliver_list(lines, out_bufsize, out_buf, err_buf);
    });
}

int fabber_get_model_outputs(void *fab, unsigned int out_bufsize, char *out_buf, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!out_buf)
        return report(FABBER_ERR_FATAL, "Output buffer is NULL", err_buf);
    return guarded(err_buf, "Error in fabber_get_model_outputs", [&]() {
        EasyLog quiet;
        std::unique_ptr<FwdModel> model(configured_model(ctx(fab), &quiet));
        vector<string> outputs, named;
        model->GetOutputs(outputs);
        for (size_t i = 0; i < outputs.size(); i++)
            if (outputs[i] != "")
                named.push_back(outputs[i]);
        return deliver_list(named, out_bufsize, out_buf, err_buf);
    });
}

int fabber_model_evaluate(void *fab, unsigned int n_params, float *params, unsigned int n_ts, float *indata,
    float *output, char *err_buf)
{
    return fabber_model_evaluate_output(fab, n_params, params, n_ts, indata, "", output, err_buf);
}

int fabber_model_evaluate_output(void *fab, unsigned int n_params, float *params, unsigned int n_ts, float *indata,
    const char *output_name, float *output, char *err_buf)
{
    if (!fab)
        return report(FABBER_ERR_FATAL, "Rundata is NULL", err_buf);
    if (!params)
        return report(FABBER_ERR_FATAL, "Params array is NULL", err_buf);
    if (!output_name)
        return report(FABBER_ERR_FATAL, "Output name is NULL", err_buf);
    if (!output)
        return report(FABBER_ERR_FATAL, "Output buffer is NULL", err_buf);
    int rc = guarded(err_buf, "Error evaluating model", [&]() {
        EasyLog log;
        stringstream sink;
        log.StartLog(sink);
        std::unique_ptr<FwdModel> model(configured_model(ctx(fab), &log));
        NEWMAT::ColumnVector p(n_params), result(n_ts), data(n_ts), coords(3);
        for (unsigned int i = 0; i < n_params; i++)
            p(i + 1) = params[i];
        for (unsigned int i = 0; i < n_ts; i++)
            data(i + 1) = indata ? indata[i] : 0;
        coords = 1;
        model->PassData(1, data, coords);
        model->EvaluateModel(p, result, output_name);
        for (unsigned int i = 0; i < n_ts; i++)
            output[i] = ((int)i < result.Nrows()) ? (float)result(i + 1) : 0.0f; // model may return fewer points
        return 0;
    });
    return rc == -1 ? FABBER_ERR_FATAL : rc;
}

} // extern "C"
