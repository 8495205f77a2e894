// rundata.cc - option store, in-memory voxel data and the run driver. See fabber_core/rundata.h.
//
// Behaviour follows the reference's FabberRunData (rundata.cc): option getters mark keys as
// used and unused keys only produce a warning (:648-658); GetVoxelData follows one level of
// key -> key indirection per step until a stored matrix is found (:802-824); the main data may
// be split over data1..n and interleaved or concatenated (:840-905); Run() = model, technique,
// DoCalculations, SaveResults (:248-311).
#include "rundata.h"

#include "fwdmodel.h"
#include "inference.h"
#include "setup.h"
#include "tools.h"
#include "version.h"

#include <chrono>
#include <errno.h>
#include <fstream>
#include <iostream>
#include <memory>
#include <sys/stat.h>
#include <time.h>

using namespace std;
using NEWMAT::Matrix;

std::ostream &operator<<(std::ostream &out, const OptionType value)
{
    static const char *names[] = { "BOOL", "STR", "INT", "FLOAT", "FILE", "IMAGE", "TIMESERIES", "MVN", "MATRIX" };
    if ((int)value >= 0 && (int)value < 9)
        return out << names[(int)value];
    return out << "UNKNOWN";
}

std::ostream &operator<<(std::ostream &out, const OptionSpec &value)
{
    out << "--" << value.name << " [" << value.type << "," << (value.optional == OPT_REQ ? "REQUIRED" : "NOT REQUIRED")
        << "," << (value.def == "" ? "NO DEFAULT" : "DEFAULT=" + value.def) << "]" << endl
        << "        " << value.description << endl;
    return out;
}

void PercentProgressCheck::Progress(int voxel, int nVoxels)
{
    int percent = (nVoxels == 0) ? 100 : (100 * voxel) / nVoxels;
    if (percent > m_last)
    {
        m_last = percent;
        cout << "\r" << percent << "%" << flush;
        if (percent == 100)
            cout << endl;
    }
}

void SimpleProgressCheck::Progress(int voxel, int nVoxels)
{
    int percent = (nVoxels == 0) ? 100 : (100 * voxel) / nVoxels;
    if (percent > m_last)
    {
        m_last = percent;
        cout << percent << endl << flush;
    }
}

static OptionSpec RUN_OPTIONS[] = {
    { "help", OPT_BOOL, "Print usage; with --method or --model show that component's options", OPT_NONREQ, "" },
    { "listmethods", OPT_BOOL, "List all known inference methods", OPT_NONREQ, "" },
    { "listmodels", OPT_BOOL, "List all known forward models", OPT_NONREQ, "" },
    { "listparams", OPT_BOOL, "List model parameters (needs the model options)", OPT_NONREQ, "" },
    { "descparams", OPT_BOOL, "Describe model parameters: name, description, units", OPT_NONREQ, "" },
    { "listoutputs", OPT_BOOL, "List additional model outputs (needs the model options)", OPT_NONREQ, "" },
    { "evaluate", OPT_STR, "Evaluate the model; value = output name or blank for the prediction", OPT_NONREQ, "" },
    { "evaluate-params", OPT_MATRIX, "Parameter values for --evaluate", OPT_NONREQ, "" },
    { "evaluate-nt", OPT_INT, "Number of time points for --evaluate", OPT_NONREQ, "" },
    { "simple-output", OPT_BOOL, "Print progress as plain percentages, one per line", OPT_NONREQ, "" },
    { "output", OPT_STR, "Directory for output files (including logfile)", OPT_REQ, "" },
    { "overwrite", OPT_BOOL, "Overwrite existing output instead of appending '+' to the directory name", OPT_NONREQ, "" },
    { "link-to-latest", OPT_BOOL, "Create a link <output>_latest to the newest output directory", OPT_NONREQ, "" },
    { "method", OPT_STR, "Use this inference method", OPT_REQ, "" },
    { "model", OPT_STR, "Use this forward model", OPT_REQ, "" },
    { "loadmodels", OPT_FILE, "Load models from this shared library", OPT_NONREQ, "" },
    { "data", OPT_TIMESERIES, "Single input data file", OPT_REQ, "" },
    { "data<n>", OPT_TIMESERIES, "Multiple data files, n=1, 2, 3...", OPT_NONREQ, "" },
    { "data-order", OPT_STR, "How multiple data files are combined: interleave, concatenate or singlefile", OPT_NONREQ,
        "interleave" },
    { "mask", OPT_IMAGE, "Mask file: inference only where mask value > 0", OPT_NONREQ, "" },
    { "mt<n>", OPT_INT, "Masked time points (1-based), ignored in the parameter updates", OPT_NONREQ, "" },
    { "suppdata", OPT_TIMESERIES, "Supplemental timeseries data required by some models", OPT_NONREQ, "" },
    { "dump-param-names", OPT_BOOL, "Write paramnames.txt with the model parameter names", OPT_NONREQ, "" },
    { "save-model-fit", OPT_BOOL, "Output the model prediction as a 4d volume", OPT_NONREQ, "" },
    { "save-residuals", OPT_BOOL, "Output data minus model prediction", OPT_NONREQ, "" },
    { "save-model-extras", OPT_BOOL, "Output additional model-specific timeseries", OPT_NONREQ, "" },
    { "save-mvn", OPT_BOOL, "Output the final MVN distributions", OPT_NONREQ, "" },
    { "save-mean", OPT_BOOL, "Output the parameter means", OPT_NONREQ, "" },
    { "save-std", OPT_BOOL, "Output the parameter standard deviations", OPT_NONREQ, "" },
    { "save-var", OPT_BOOL, "Output the parameter variances", OPT_NONREQ, "" },
    { "save-zstat", OPT_BOOL, "Output the parameter Z-statistics", OPT_NONREQ, "" },
    { "save-noise-mean", OPT_BOOL, "Output the noise (precision) means", OPT_NONREQ, "" },
    { "save-noise-std", OPT_BOOL, "Output the noise (precision) standard deviations", OPT_NONREQ, "" },
    { "save-free-energy", OPT_BOOL, "Output the free energy, if calculated", OPT_NONREQ, "" },
    { "optfile", OPT_BOOL, "File with additional options, one per line, as on the command line", OPT_NONREQ, "" },
    { "debug", OPT_BOOL, "Very verbose logging; only for tiny numbers of voxels", OPT_NONREQ, "" },
    { "device", OPT_INT, "MI355X: index of the GPU to run on", OPT_NONREQ, "0" },
    { "devices", OPT_STR, "MI355X: 'all' or a comma-separated list of GPU indices; voxelwise VB shards the voxels over them", OPT_NONREQ, "" },
    { "" },
};

void FabberRunData::GetOptions(std::vector<OptionSpec> &opts)
{
    for (int i = 0; RUN_OPTIONS[i].name != ""; i++)
        opts.push_back(RUN_OPTIONS[i]);
}

FabberRunData::FabberRunData(bool compat_options)
    : m_progress(0)
{
    init(compat_options);
}

void FabberRunData::init(bool compat_options)
{
    FabberSetup::SetupDefaults();
    if (compat_options && !GetBool("no-compat-output"))
    {
        static const char *defaults[]
            = { "save-mean", "save-std", "save-zstat", "save-noise-mean", "save-noise-std", "save-free-energy", "save-mvn" };
        for (size_t i = 0; i < sizeof(defaults) / sizeof(defaults[0]); i++)
            SetBool(defaults[i]);
    }
}

FabberRunData::~FabberRunData()
{
}

void FabberRunData::LogParams()
{
    for (map<string, string>::iterator it = m_params.begin(); it != m_params.end(); ++it)
        LOG << "FabberRunData::Parameter " << it->first << "=" << it->second << endl;
}

void FabberRunData::Run(ProgressCheck *progress)
{
    if (!m_log)
        m_log = &m_default_log;
    m_progress = progress;

    LOG << "FabberRunData::FABBER release: " << fabber_version() << endl;
    time_t start;
    time(&start);
    LOG << "FabberRunData::Start time: " << ctime(&start);
    LogParams();

    std::unique_ptr<FwdModel> fwd_model(FwdModel::NewFromName(GetString("model")));
    fwd_model->SetLogger(m_log);
    fwd_model->Initialize(*this);

    std::vector<Parameter> params;
    fwd_model->GetParameters(*this, params);
    if (params.empty())
        throw FabberInternalError("Model has no parameters");
    LOG << "FabberRunData::Forward Model version " << fwd_model->ModelVersion() << endl;

    if (GetBool("dump-param-names"))
    {
        ofstream names((GetStringDefault("output", ".") + "/paramnames.txt").c_str());
        for (size_t i = 0; i < params.size(); i++)
            names << params[i].name << endl;
    }

    // FVB_HOST_TIMING: where the host side of a run spends its time (stderr)
    const bool timing = getenv("FVB_HOST_TIMING") != NULL;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    std::unique_ptr<InferenceTechnique> infer(InferenceTechnique::NewFromName(GetString("method")));
    infer->Initialize(fwd_model.get(), *this);
    const auto t1 = now();

    int nvoxels = GetVoxelCoords().Ncols();
    LOG << "FabberRunData::Num voxels " << nvoxels << endl;
    Progress(0, nvoxels);
    infer->DoCalculations(*this);
    const auto t2 = now();
    Progress(nvoxels, nvoxels);
    LOG << "FabberRunData::Saving results " << endl;
    infer->SaveResults(*this);
    const auto t3 = now();
    LOG << "FabberRunData::All done." << endl;
    if (timing)
        fprintf(stderr, "[fabber host] Initialize %.1f ms, DoCalculations %.1f ms, SaveResults %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3));

    CheckAllOptionsUsed();
    time_t end;
    time(&end);
    LOG << "FabberRunData::Start time: " << ctime(&start);
    LOG << "FabberRunData::End time: " << ctime(&end);
    LOG << "FabberRunData::Duration: " << end - start << " seconds." << endl;
}

// ---------------------------------------------------------------------------------------------
// Option parsing
// ---------------------------------------------------------------------------------------------
static string strip(const string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == string::npos)
        return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

void FabberRunData::AddKeyEqualsValue(const string &exp, bool trim_comments)
{
    string key = exp;
    if (trim_comments)
    {
        size_t hash = key.find('#');
        if (hash != string::npos)
            key.erase(hash);
    }
    key = strip(key);
    if (key.empty())
        return;
    string value;
    size_t eq = key.find('=');
    if (eq != string::npos)
    {
        value = strip(key.substr(eq + 1));
        key = strip(key.substr(0, eq));
        // allow quoted values
        if (value.size() >= 2 && ((value[0] == '"' && value[value.size() - 1] == '"') || (value[0] == '\'' && value[value.size() - 1] == '\'')))
            value = value.substr(1, value.size() - 2);
    }
    if (m_params.count(key) > 0)
        throw InvalidOptionValue(key, value, "Already has a value: " + m_params[key]);
    m_params[key] = value;
}

void FabberRunData::ParseParamFile(const string &filename)
{
    ifstream in(filename.c_str());
    if (!in.good())
        throw InvalidOptionValue("-f", filename, "Could not open option file");
    string line;
    while (getline(in, line))
    {
        string s = strip(line);
        if (s.compare(0, 2, "--") == 0)
            s = s.substr(2);
        AddKeyEqualsValue(s, true);
    }
}

void FabberRunData::ParseOldStyleParamFile(const string &filename)
{
    // whitespace-separated --key=value tokens, '#' to end of line is a comment
    ifstream in(filename.c_str());
    if (!in.good())
        throw InvalidOptionValue("-@", filename, "Could not open option file");
    string line;
    while (getline(in, line))
    {
        size_t hash = line.find('#');
        if (hash != string::npos)
            line.erase(hash);
        istringstream tokens(line);
        string tok;
        while (tokens >> tok)
        {
            if (tok.compare(0, 2, "--") != 0)
                throw InvalidOptionValue(filename, tok, "Options must begin with --");
            AddKeyEqualsValue(tok.substr(2));
        }
    }
}

void FabberRunData::Parse(int argc, char **argv)
{
    m_params[""] = argv[0];
    for (int a = 1; a < argc; a++)
    {
        string arg = argv[a];
        if (arg == "-f")
        {
            if (++a >= argc)
                throw InvalidOptionValue("-f", "", "Needs a filename");
            ParseParamFile(argv[a]);
        }
        else if (arg == "-@")
        {
            if (++a >= argc)
                throw InvalidOptionValue("-@", "", "Needs a filename");
            ParseOldStyleParamFile(argv[a]);
        }
        else if (arg.compare(0, 2, "--") == 0)
        {
            string body = arg.substr(2);
            if (body.find('=') == string::npos && (body == "optfile") && a + 1 < argc)
                body += string("=") + argv[++a];
            AddKeyEqualsValue(body);
        }
        else
        {
            throw InvalidOptionValue(arg, "", "Options must begin with --");
        }
    }
    if (m_params.count("optfile"))
    {
        string f = m_params["optfile"];
        m_params.erase("optfile");
        ParseParamFile(f);
    }
}

void FabberRunData::Set(const string &key, const string &value)
{
    m_params[key] = value;
}
void FabberRunData::Set(const string &key, double value)
{
    m_params[key] = stringify(value);
}
void FabberRunData::SetBool(const string &key, bool value)
{
    if (value)
        m_params[key] = "";
    else
        m_params.erase(key);
}
void FabberRunData::Unset(const std::string &key)
{
    m_params.erase(key);
}
bool FabberRunData::HaveKey(const string &key)
{
    return m_params.count(key) > 0;
}

string FabberRunData::GetString(const string &key)
{
    return Read(key, key);
}

string FabberRunData::GetStringDefault(const string &key, const string &def) const
{
    m_used_params.insert(key);
    map<string, string>::const_iterator it = m_params.find(key);
    return (it == m_params.end()) ? def : it->second;
}

std::vector<std::string> FabberRunData::GetStringList(const std::string &prefix)
{
    std::vector<std::string> out;
    if (HaveKey(prefix))
        out.push_back(GetString(prefix));
    else
        for (int n = 1; HaveKey(prefix + stringify(n)); n++)
            out.push_back(GetString(prefix + stringify(n)));
    return out;
}

bool FabberRunData::GetBool(const string &key)
{
    map<string, string>::iterator it = m_params.find(key);
    if (it == m_params.end())
        return false;
    m_used_params.insert(key);
    if (it->second == "")
        return true;
    throw InvalidOptionValue(key, it->second, "Value should not be given for boolean option");
}

int FabberRunData::GetInt(const string &key, int min, int max)
{
    string val = GetString(key);
    int i;
    try
    {
        i = convertTo<int>(val, key);
    }
    catch (InvalidOptionValue &)
    {
        throw InvalidOptionValue(key, val, "Must be an integer");
    }
    if (i < min)
        throw InvalidOptionValue(key, val, "Minimum " + stringify(min));
    if (i > max)
        throw InvalidOptionValue(key, val, "Maximum " + stringify(max));
    return i;
}
int FabberRunData::GetIntDefault(const string &key, int def, int min, int max)
{
    return HaveKey(key) ? GetInt(key, min, max) : def;
}
std::vector<int> FabberRunData::GetIntList(const std::string &prefix, int min, int max)
{
    std::vector<int> out;
    if (HaveKey(prefix))
        out.push_back(GetInt(prefix, min, max));
    else
        for (int n = 1; HaveKey(prefix + stringify(n)); n++)
            out.push_back(GetInt(prefix + stringify(n), min, max));
    return out;
}

double FabberRunData::GetDouble(const string &key, double min, double max)
{
    string val = GetString(key);
    double d;
    try
    {
        d = convertTo<double>(val, key);
    }
    catch (InvalidOptionValue &)
    {
        throw InvalidOptionValue(key, val, "Must be an number");
    }
    if (d < min)
        throw InvalidOptionValue(key, val, "Minimum " + stringify(min));
    if (d > max)
        throw InvalidOptionValue(key, val, "Maximum " + stringify(max));
    return d;
}
double FabberRunData::GetDoubleDefault(const string &key, double def, double min, double max)
{
    return HaveKey(key) ? GetDouble(key, min, max) : def;
}
std::vector<double> FabberRunData::GetDoubleList(const std::string &prefix, double min, double max)
{
    std::vector<double> out;
    if (HaveKey(prefix))
        out.push_back(GetDouble(prefix, min, max));
    else
        for (int n = 1; HaveKey(prefix + stringify(n)); n++)
            out.push_back(GetDouble(prefix + stringify(n), min, max));
    return out;
}

string FabberRunData::Read(const string &key, const string &msg)
{
    map<string, string>::iterator it = m_params.find(key);
    if (it == m_params.end())
        throw MandatoryOptionMissing(msg);
    if (it->second == "")
        throw InvalidOptionValue(key, "<no value>", "Value must be given");
    m_used_params.insert(key);
    return it->second;
}
std::string FabberRunData::Read(const std::string &key)
{
    return GetString(key);
}
std::string FabberRunData::ReadWithDefault(const std::string &key, const std::string &def)
{
    return GetStringDefault(key, def);
}
bool FabberRunData::ReadBool(const std::string &key)
{
    return GetBool(key);
}

void FabberRunData::CheckAllOptionsUsed() const
{
    for (map<string, string>::const_iterator it = m_params.begin(); it != m_params.end(); ++it)
        if (it->first != "" && m_used_params.count(it->first) == 0)
            WARN_ONCE("Unused option specified: " + it->first);
}

string FabberRunData::GetOutputDir()
{
    GetBool("link-to-latest");
    if (m_outdir != "")
        return m_outdir;
    string base = GetStringDefault("output", "");
    if (base == "")
        return m_outdir = ".";
    bool overwrite = GetBool("overwrite");
    m_outdir = base;
    for (int tries = 0; tries < 50; tries++)
    {
        errno = 0;
        if (mkdir(m_outdir.c_str(), 0777) == 0)
            return m_outdir;
        if (errno != EEXIST)
            break;
        if (overwrite)
            return m_outdir;
        m_outdir += "+";
    }
    throw FabberInternalError("Cannot create output directory (bad path, or too many + signs?): " + m_outdir);
}

ostream &operator<<(ostream &out, const FabberRunData &opts)
{
    for (map<string, string>::const_iterator i = opts.m_params.begin(); i != opts.m_params.end(); ++i)
    {
        if (i->second == "")
            out << "--" << i->first << endl;
        else
            out << "--" << i->first << "='" << i->second << "'" << endl;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// Voxel data
// ---------------------------------------------------------------------------------------------
const Matrix &FabberRunData::GetMainVoxelData()
{
    try
    {
        return GetVoxelData("data");
    }
    catch (DataNotFound &first)
    {
        try
        {
            GetVoxelData("data1");
        }
        catch (DataNotFound &)
        {
            throw first;
        }
        return GetMainVoxelDataMultiple();
    }
}

const Matrix &FabberRunData::GetVoxelSuppData()
{
    try
    {
        return GetVoxelData("suppdata");
    }
    catch (DataNotFound &)
    {
        return m_empty;
    }
}

int FabberRunData::GetVoxelDataSize(const std::string &key)
{
    return GetVoxelData(key).Nrows();
}

const Matrix &FabberRunData::GetVoxelCoords()
{
    return GetVoxelData("coords");
}

const Matrix &FabberRunData::GetVoxelData(const std::string &key)
{
    // An option may name the data key that holds the matrix (e.g. continue-from-mvn=mvns);
    // follow such references until there is no further option of that name.
    string cur = key, data_key;
    while (cur != "")
    {
        data_key = cur;
        cur = GetStringDefault(cur, "");
        if (cur == key)
            break; // circular reference
    }
    return LoadVoxelData(data_key);
}

const Matrix &FabberRunData::LoadVoxelData(const std::string &key)
{
    map<string, Matrix>::iterator it = m_voxel_data.find(key);
    if (it == m_voxel_data.end())
    {
        // held as float32 (SetVoxelDataF32): somebody wants the matrix now
        map<string, F32Image>::iterator f = m_voxel_data_f32.find(key);
        if (f == m_voxel_data_f32.end())
            throw DataNotFound(key);
        const int rows = f->second.rows;
        const size_t cols = rows > 0 ? f->second.values.size() / (size_t)rows : 0;
        Matrix &m = m_voxel_data[key];
        m.ReSizeNoInit(rows, (int)cols);
        const float *src = f->second.values.data();
        double *dst = m.Store();
        fabber_parallel_for(rows, [&](int r) {
            const float *s = src + (size_t)r * cols;
            double *d = dst + (size_t)r * cols;
            for (size_t i = 0; i < cols; i++)
                d[i] = s[i];
        });
        return m;
    }
    return it->second;
}

void FabberRunData::SetVoxelDataF32(string key, int rows, FabberF32Values &&values)
{
    const size_t cols = rows > 0 ? values.size() / (size_t)rows : 0;
    map<string, Matrix>::iterator coords = m_voxel_data.find("coords");
    if (coords != m_voxel_data.end() && key != "coords" && (int)cols != coords->second.Ncols())
        throw InvalidOptionValue(key, stringify(cols) + " voxels",
            "Number of voxels does not match the co-ordinates (" + stringify(coords->second.Ncols()) + ")");
    m_voxel_data.erase(key);
    F32Image &img = m_voxel_data_f32[key];
    img.rows = rows;
    img.values = std::move(values);
}

const float *FabberRunData::GetMainVoxelDataF32(int &rows, int &cols)
{
    // (the key "data" itself, not a reference to another key, and no matrix of that name: GetMainVoxelData would
    // find the same series)
    if (GetStringDefault("data", "") != "" || m_voxel_data.find("data") != m_voxel_data.end())
        return NULL;
    map<string, F32Image>::iterator f = m_voxel_data_f32.find("data");
    if (f == m_voxel_data_f32.end() || f->second.rows <= 0)
        return NULL;
    rows = f->second.rows;
    cols = (int)(f->second.values.size() / (size_t)rows);
    return f->second.values.data();
}

const Matrix &FabberRunData::GetMainVoxelDataMultiple()
{
    vector<Matrix> sets;
    for (int n = 1;; n++)
    {
        try
        {
            sets.push_back(GetVoxelData("data" + stringify(n)));
        }
        catch (DataNotFound &)
        {
            break;
        }
    }
    string order = GetStringDefault("data-order", "interleave");
    const int nSets = (int)sets.size();
    if (nSets < 1)
        throw DataNotFound("data");
    if (order == "singlefile" && nSets > 1)
        throw InvalidOptionValue("data-order", "singlefile", "More than one file specified");

    if (order == "interleave")
    {
        const int nTimes = sets[0].Nrows();
        m_mainDataMultiple.ReSize(nTimes * nSets, sets[0].Ncols());
        for (int j = 0; j < nSets; j++)
        {
            if (sets[j].Nrows() != nTimes)
                throw InvalidOptionValue(
                    "data-order", "interleave", "Data sets must all have the same number of time points");
            for (int i = 0; i < nTimes; i++)
                m_mainDataMultiple.Row(nSets * i + j + 1) = sets[j].Row(i + 1);
        }
    }
    else if (order == "concatenate")
    {
        m_mainDataMultiple = sets[0];
        for (int j = 1; j < nSets; j++)
            m_mainDataMultiple &= sets[j];
    }
    else if (order == "singlefile")
    {
        m_mainDataMultiple = sets[0];
    }
    else
    {
        throw InvalidOptionValue("data-order", order, "Value not recognized");
    }
    LOG << "FabberRunData::Done loading data, size = " << m_mainDataMultiple.Nrows() << " timepoints by "
        << m_mainDataMultiple.Ncols() << " voxels" << endl;
    return m_mainDataMultiple;
}

void FabberRunData::ClearVoxelData(string key)
{
    if (key != "")
    {
        m_voxel_data.erase(key);
        m_voxel_data_f32.erase(key);
    }
    else
    {
        m_voxel_data.clear();
        m_voxel_data_f32.clear();
    }
}

void FabberRunData::CheckSize(std::string key, const Matrix &mat)
{
    // every data item must have one column per voxel
    map<string, Matrix>::iterator coords = m_voxel_data.find("coords");
    if (coords != m_voxel_data.end() && key != "coords" && mat.Ncols() != coords->second.Ncols())
        throw InvalidOptionValue(key, stringify(mat.Ncols()) + " voxels",
            "Number of voxels does not match the co-ordinates (" + stringify(coords->second.Ncols()) + ")");
}

void FabberRunData::SetVoxelData(string key, const Matrix &data)
{
    CheckSize(key, data);
    m_voxel_data_f32.erase(key);
    m_voxel_data[key] = data;
}

void FabberRunData::SaveVoxelData(const std::string &filename, Matrix &data, VoxelDataType)
{
    LOG << "FabberRunData::Saving to memory: " << filename << endl;
    SetVoxelData(filename, data);
}

void FabberRunData::SetVoxelCoords(const Matrix &coords)
{
    if (coords.Ncols() > 0 && coords.Nrows() != 3)
        throw InvalidOptionValue(
            "Coordinates dimensions", stringify(coords.Nrows()), "Co-ordinates must be 3 dimensional");
    m_voxel_data["coords"] = coords;
    if (m_extent.empty())
    {
        m_extent.assign(3, 0);
        m_dims.assign(3, 1.0f);
        for (int d = 0; d < 3; d++)
            if (coords.Ncols() > 0)
                m_extent[d] = (int)(coords.Row(d + 1).Maximum() - coords.Row(d + 1).Minimum()) + 1;
    }
}

void FabberRunData::GetExtent(std::vector<int> &extent, std::vector<float> &dims)
{
    extent = m_extent;
    dims = m_dims;
}

void FabberRunData::SetExtent(int nx, int ny, int nz, float sx, float sy, float sz)
{
    if (nx < 0 || ny < 0 || nz < 0)
        throw FabberRunDataError("Extent must be non-negative");
    m_extent.resize(3);
    m_extent[0] = nx;
    m_extent[1] = ny;
    m_extent[2] = nz;
    m_dims.resize(3);
    m_dims[0] = sx;
    m_dims[1] = sy;
    m_dims[2] = sz;
}
