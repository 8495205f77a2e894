/* fabber_core.cc - the `fabber` command line tool on the MI355X engine.
 *
 * Behaviour of the reference's execute() (fabber_core.cc:88-323): options with --key[=value] or
 * -f <file>; the information queries (--help, --version, --listmodels, --listmethods,
 * --listparams, --descparams, --listoutputs, --evaluate) answer on stdout without touching any
 * data; otherwise data / mask are NIfTI files, results go to --output as NIfTI plus logfile and
 * paramnames.txt. Exit code 0 on success, 1 after any exception (message on stderr and in the log). */
#include "fabber_core/fabber_core.h"

#include "fabber_core/fwdmodel.h"
#include "fabber_core/inference.h"
#include "fabber_core/rundata_newimage.h"
#include "fabber_core/tools.h"
#include "fabber_core/version.h"

#include "armawrap/newmat.h"

#include <exception>
#include <iostream>
#include <memory>
#include <stdlib.h>
#include <string>
#include <vector>

using namespace std;
using NEWMAT::ColumnVector;
using NEWMAT::Matrix;

static void Version()
{
    cout << "Fabber " << fabber_version() << " : " << fabber_source_date() << endl;
}

static void Usage()
{
    Version();
    cout << "Usage: fabber [--<option>|--<option>=<value> ...]" << endl
         << endl
         << "Use -f <file> to read options in option=value form" << endl
         << "Use -@ <file> to read options in command line form (DEPRECATED)." << endl
         << endl
         << "General options " << endl
         << endl;
    vector<OptionSpec> options;
    FabberRunData::GetOptions(options);
    for (unsigned int i = 0; i < options.size(); i++)
        cout << options[i] << endl;
}

/** A model set up from the options, with its log output swallowed */
static std::unique_ptr<FwdModel> quiet_model(FabberRunData &params, EasyLog &sink)
{
    std::unique_ptr<FwdModel> fwd_model(FwdModel::NewFromName(params.GetStringDefault("model", "")));
    fwd_model->SetLogger(&sink);
    fwd_model->Initialize(params);
    return fwd_model;
}

int execute(int argc, char **argv)
{
    EasyLog log;
    bool gzLog = false;
    bool simple_output = false;
    int ret = 1;

    try
    {
        setenv("FSLOUTPUTTYPE", "NIFTI_GZ", 0); // may be missing if FSL is not installed

        FabberRunDataNewimage paramso(true);
        FabberRunDataNewimage *params = &paramso;
        params->SetLogger(&log);
        params->Parse(argc, argv);

        if (params->GetBool("help") || argc == 1)
        {
            string model = params->GetStringDefault("model", "");
            string method = params->GetStringDefault("method", "");
            if (model != "")
                FwdModel::UsageFromName(model, cout);
            else if (method != "")
                InferenceTechnique::UsageFromName(method, cout);
            else
                Usage();
            return 0;
        }
        if (params->GetBool("version"))
        {
            string model_name = params->GetStringDefault("model", "");
            if (model_name != "")
            {
                std::unique_ptr<FwdModel> model(FwdModel::NewFromName(model_name));
                cout << model->ModelVersion() << endl;
            }
            else
                Version();
            return 0;
        }
        if (params->GetBool("listmodels"))
        {
            vector<string> models = FwdModel::GetKnown();
            for (size_t i = 0; i < models.size(); i++)
                cout << models[i] << endl;
            return 0;
        }
        if (params->GetBool("listmethods"))
        {
            vector<string> infers = InferenceTechnique::GetKnown();
            for (size_t i = 0; i < infers.size(); i++)
                cout << infers[i] << endl;
            return 0;
        }
        if (params->GetBool("listparams") || params->GetBool("descparams"))
        {
            const bool describe = params->GetBool("descparams");
            EasyLog sink;
            std::unique_ptr<FwdModel> fwd_model = quiet_model(*params, sink);
            vector<Parameter> model_params;
            fwd_model->GetParameters(*params, model_params);
            for (size_t i = 0; i < model_params.size(); i++)
            {
                cout << model_params[i].name;
                if (describe)
                {
                    cout << " " << model_params[i].desc;
                    if (model_params[i].units != "")
                        cout << " (units: " << model_params[i].units << ")";
                }
                cout << endl;
            }
            return 0;
        }
        if (params->GetBool("listoutputs"))
        {
            EasyLog sink;
            std::unique_ptr<FwdModel> fwd_model = quiet_model(*params, sink);
            vector<string> model_outputs;
            fwd_model->GetOutputs(model_outputs);
            for (size_t i = 0; i < model_outputs.size(); i++)
                cout << model_outputs[i] << endl;
            return 0;
        }
        if (params->HaveKey("evaluate"))
        {
            EasyLog sink;
            std::unique_ptr<FwdModel> fwd_model = quiet_model(*params, sink);
            Matrix param_values = fabber::read_matrix_file(params->GetString("evaluate-params"));
            ColumnVector p_vec = param_values.Column(1);
            int n_ts = params->GetInt("evaluate-nt", 0);
            ColumnVector data_vec(n_ts);
            for (int i = 1; i <= n_ts; i++)
                data_vec(i) = 0;
            if (params->HaveKey("evaluate-data"))
            {
                Matrix data_values = fabber::read_matrix_file(params->GetString("evaluate-data"));
                data_vec = data_values.Column(1);
            }
            ColumnVector coords(3);
            coords(1) = coords(2) = coords(3) = 1;
            fwd_model->PassData(1, data_vec, coords);
            ColumnVector o_vec(n_ts);
            fwd_model->EvaluateModel(p_vec, o_vec, params->GetStringDefault("evaluate", ""));
            for (int i = 0; i < o_vec.Nrows(); i++)
                cout << o_vec(i + 1) << endl;
            return 0;
        }

        params->SetBool("dump-param-names"); // the command line tool writes paramnames.txt
        params->SetBool("link-to-latest");
        params->SetExtentFromData();
        simple_output = params->GetBool("simple-output");

        log.StartLog(params->GetOutputDir());
        if (!simple_output)
        {
            cout << "----------------------" << endl;
            cout << "Welcome to FABBER " << fabber_version() << endl;
            cout << "----------------------" << endl;
            cout << "Last commit: " << fabber_source_date() << endl;
            cout << "Logfile started: " << log.GetOutputDirectory() << "/logfile" << endl;
            PercentProgressCheck progress;
            params->Run(&progress);
        }
        else
        {
            SimpleProgressCheck progress;
            params->Run(&progress);
        }
        log.ReissueWarnings();
        gzLog = params->GetBool("gzip-log"); // only gzip the log if we exit normally
        ret = 0;
    }
    catch (NEWMAT::Exception &e)
    {
        log.ReissueWarnings();
        log.LogStream() << "NEWMAT exception caught in fabber:\n  " << e.what() << endl;
        cerr << "NEWMAT exception caught in fabber:\n  " << e.what() << endl;
    }
    catch (const exception &e)
    {
        log.ReissueWarnings();
        log.LogStream() << "Exception caught in fabber:\n  " << e.what() << endl;
        cerr << "Exception caught in fabber:\n  " << e.what() << endl;
    }
    catch (...)
    {
        log.ReissueWarnings();
        log.LogStream() << "Some other exception caught in fabber!" << endl;
        cerr << "Some other exception caught in fabber!" << endl;
    }

    if (log.LogStarted())
    {
        if (!simple_output)
            cout << endl << "Final logfile: " << log.GetOutputDirectory() << (gzLog ? "/logfile.gz" : "/logfile") << endl;
        log.StopLog(gzLog);
    }
    else
    {
        log.StartLog(cerr); // never got as far as the logfile: flush what was buffered to stderr
        log.StopLog();
    }
    return ret;
}
