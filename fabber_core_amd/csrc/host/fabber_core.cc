/* fabber_core.cc - the `fabber` command line tool on the MI355X engine.
 *
 * What the tool does (the behaviour of the reference's execute(), fabber_core.cc:88-323, and of
 * tests/test_cli_nifti.py):
 *   - options come as --key, --key=value or from "-f <file>" (one key=value per line);
 *   - a set of "questions" is answered on stdout without touching any image: no arguments / --help
 *     (general usage, or that of --model / --method), --version (of the tool or of --model),
 *     --listmodels, --listmethods, --listparams / --descparams, --listoutputs, --evaluate[=<output>]
 *     with --evaluate-params, --evaluate-nt and optionally --evaluate-data;
 *   - anything else is a fit: data and mask are NIfTI files, the result images, the logfile and
 *     paramnames.txt go to --output (made unique with '+' signs, "<output>_latest" links to it);
 *   - exit code 0 on success, 1 after any exception; the message goes to stderr and to the logfile, or,
 *     when the failure came before the logfile existed, everything logged so far goes to stderr.
 *
 * Structure: the questions are rows of one table (flag -> predicate -> answer); execute() parses,
 * walks the table and otherwise hands over to fit(); Session owns the log and the exit bookkeeping.
 */
#include "fabber_core/fabber_core.h"

#include "fabber_core/fwdmodel.h"
#include "fabber_core/inference.h"
#include "fabber_core/rundata_newimage.h"
#include "fabber_core/tools.h"
#include "fabber_core/version.h"

#include "armawrap/newmat.h"

#include <cstdlib>
#include <exception>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace
{
typedef FabberRunDataNewimage Options;

void print_lines(const std::vector<std::string> &lines)
{
    for (const std::string &l : lines)
        std::cout << l << std::endl;
}

void print_banner_version()
{
    std::cout << "Fabber " << fabber_version() << " : " << fabber_source_date() << std::endl;
}

/** The model named by --model, initialised from the options, talking to a log nobody reads. */
struct SilentModel
{
    EasyLog sink;
    std::unique_ptr<FwdModel> model;
    explicit SilentModel(Options &opts)
        : model(FwdModel::NewFromName(opts.GetStringDefault("model", "")))
    {
        model->SetLogger(&sink);
        model->Initialize(opts);
    }
    FwdModel *operator->()
    {
        return model.get();
    }
};

// ---- the questions ---------------------------------------------------------------------------------
void answer_help(Options &opts)
{
    const std::string model = opts.GetStringDefault("model", ""), method = opts.GetStringDefault("method", "");
    if (!model.empty())
    {
        FwdModel::UsageFromName(model, std::cout);
        return;
    }
    if (!method.empty())
    {
        InferenceTechnique::UsageFromName(method, std::cout);
        return;
    }
    print_banner_version();
    std::cout << "Usage: fabber [--<option>|--<option>=<value> ...]\n\n"
              << "Use -f <file> to read options in option=value form\n"
              << "Use -@ <file> to read options in command line form (DEPRECATED).\n\n"
              << "General options \n"
              << std::endl;
    std::vector<OptionSpec> general;
    FabberRunData::GetOptions(general);
    for (const OptionSpec &o : general)
        std::cout << o << std::endl;
}

void answer_version(Options &opts)
{
    const std::string model = opts.GetStringDefault("model", "");
    if (model.empty())
        print_banner_version();
    else
        std::cout << std::unique_ptr<FwdModel>(FwdModel::NewFromName(model))->ModelVersion() << std::endl;
}

void answer_models(Options &)
{
    print_lines(FwdModel::GetKnown());
}

void answer_methods(Options &)
{
    print_lines(InferenceTechnique::GetKnown());
}

void answer_params(Options &opts)
{
    SilentModel m(opts);
    std::vector<Parameter> ps;
    m->GetParameters(opts, ps);
    const bool with_description = opts.GetBool("descparams");
    for (const Parameter &p : ps)
    {
        std::cout << p.name;
        if (with_description)
        {
            std::cout << " " << p.desc;
            if (!p.units.empty())
                std::cout << " (units: " << p.units << ")";
        }
        std::cout << std::endl;
    }
}

void answer_outputs(Options &opts)
{
    SilentModel m(opts);
    std::vector<std::string> names;
    m->GetOutputs(names);
    print_lines(names);
}

/** --evaluate: one model evaluation for the parameter values in a matrix file */
void answer_evaluate(Options &opts)
{
    SilentModel m(opts);
    const NEWMAT::ColumnVector values = fabber::read_matrix_file(opts.GetString("evaluate-params")).Column(1);
    const int n = opts.GetInt("evaluate-nt", 0);
    NEWMAT::ColumnVector series(n);
    series = 0.0;
    if (opts.HaveKey("evaluate-data"))
        series = fabber::read_matrix_file(opts.GetString("evaluate-data")).Column(1);
    NEWMAT::ColumnVector where(3);
    where = 1.0;
    m->PassData(1, series, where);
    NEWMAT::ColumnVector prediction(n);
    m->EvaluateModel(values, prediction, opts.GetStringDefault("evaluate", ""));
    for (int t = 1; t <= prediction.Nrows(); t++)
        std::cout << prediction(t) << std::endl;
}

struct Question
{
    bool (*asked)(Options &, int argc);
    void (*answer)(Options &);
};

template <const char *const &FLAG>
bool flag_set(Options &o, int)
{
    return o.GetBool(FLAG);
}
const char *const F_VERSION = "version", *const F_MODELS = "listmodels", *const F_METHODS = "listmethods",
                  *const F_OUTPUTS = "listoutputs";

// in the order they take precedence
const Question QUESTIONS[] = {
    { [](Options &o, int argc) { return argc == 1 || o.GetBool("help"); }, answer_help },
    { flag_set<F_VERSION>, answer_version },
    { flag_set<F_MODELS>, answer_models },
    { flag_set<F_METHODS>, answer_methods },
    { [](Options &o, int) { return o.GetBool("listparams") || o.GetBool("descparams"); }, answer_params },
    { flag_set<F_OUTPUTS>, answer_outputs },
    { [](Options &o, int) { return o.HaveKey("evaluate"); }, answer_evaluate },
};

// ---- a fit -----------------------------------------------------------------------------------------
/** The log of one invocation and what has to happen to it whichever way the invocation ends. */
class Session
{
public:
    EasyLog log;
    bool quiet = false;    // --simple-output
    bool compress = false; // --gzip-log, honoured only after a clean finish

    void failed(const std::string &headline, const char *detail)
    {
        log.ReissueWarnings();
        for (std::ostream *to : { &log.LogStream(), static_cast<std::ostream *>(&std::cerr) })
        {
            *to << headline;
            if (detail)
                *to << "\n  " << detail;
            *to << std::endl;
        }
    }

    void close()
    {
        if (!log.LogStarted())
        {
            // nothing was ever written to a file: what was buffered goes to stderr
            log.StartLog(std::cerr);
            log.StopLog();
            return;
        }
        if (!quiet)
            std::cout << std::endl
                      << "Final logfile: " << log.GetOutputDirectory() << (compress ? "/logfile.gz" : "/logfile") << std::endl;
        log.StopLog(compress);
    }
};

void fit(Options &opts, Session &s)
{
    opts.SetBool("dump-param-names"); // paramnames.txt is part of the tool's output
    opts.SetBool("link-to-latest");
    opts.SetExtentFromData();
    s.quiet = opts.GetBool("simple-output");
    s.log.StartLog(opts.GetOutputDir());
    if (s.quiet)
    {
        SimpleProgressCheck progress;
        opts.Run(&progress);
    }
    else
    {
        const std::string rule(22, '-');
        std::cout << rule << "\nWelcome to FABBER " << fabber_version() << "\n" << rule << std::endl;
        std::cout << "Last commit: " << fabber_source_date() << std::endl;
        std::cout << "Logfile started: " << s.log.GetOutputDirectory() << "/logfile" << std::endl;
        PercentProgressCheck progress;
        opts.Run(&progress);
    }
    s.log.ReissueWarnings();
    s.compress = opts.GetBool("gzip-log");
}
} // namespace

int execute(int argc, char **argv)
{
    Session session;
    int exit_code = 1;
    try
    {
        setenv("FSLOUTPUTTYPE", "NIFTI_GZ", 0); // image writers look at it; absent without an FSL installation
        Options opts(true);
        opts.SetLogger(&session.log);
        opts.Parse(argc, argv);
        for (const Question &q : QUESTIONS)
            if (q.asked(opts, argc))
            {
                q.answer(opts);
                return 0;
            }
        fit(opts, session);
        exit_code = 0;
    }
    catch (NEWMAT::Exception &e)
    {
        session.failed("NEWMAT exception caught in fabber:", e.what());
    }
    catch (const std::exception &e)
    {
        session.failed("Exception caught in fabber:", e.what());
    }
    catch (...)
    {
        session.failed("Some other exception caught in fabber!", nullptr);
    }
    session.close();
    return exit_code;
}
