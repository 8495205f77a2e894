/* rundata.h - option store + in-memory voxel data of one fabber run.
 *
 * Mirrors the public surface of the reference's FabberRunData (rundata.h:171-673): typed option
 * getters with "used" tracking, named voxel-data matrices (rows = values, columns = voxels),
 * progress callbacks, the exception family used as control flow (rundata.h:676-758) and
 * OptionSpec. Run() drives model -> inference technique -> SaveResults like rundata.cc:248-311;
 * the inference technique is where the HIP engine is entered. */
#pragma once

#include "easylog.h"

#include "armawrap/newmat.h"

#include <float.h>
#include <limits.h>
#include <map>
#include <ostream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#define DEPRECATED 7

enum OptionType
{
    OPT_BOOL,
    OPT_STR,
    OPT_INT,
    OPT_FLOAT,
    OPT_FILE,
    OPT_IMAGE,
    OPT_TIMESERIES,
    OPT_MVN,
    OPT_MATRIX
};
enum OptionReq
{
    OPT_REQ = 0,
    OPT_NONREQ = 1
};
std::ostream &operator<<(std::ostream &out, const OptionType value);

struct OptionSpec
{
    std::string name;
    OptionType type;
    std::string description;
    OptionReq optional;
    std::string def;
};
std::ostream &operator<<(std::ostream &out, const OptionSpec &value);

enum VoxelDataType
{
    VDT_SCALAR,
    VDT_MVN
};

class ProgressCheck
{
public:
    virtual ~ProgressCheck()
    {
    }
    virtual void Progress(int voxel, int nVoxels)
    {
    }
};
class PercentProgressCheck : public ProgressCheck
{
public:
    PercentProgressCheck()
        : m_last(-1)
    {
    }
    void Progress(int voxel, int nVoxels);

private:
    int m_last;
};
class SimpleProgressCheck : public ProgressCheck
{
public:
    SimpleProgressCheck()
        : m_last(-1)
    {
    }
    void Progress(int voxel, int nVoxels);

private:
    int m_last;
};
class CallbackProgressCheck : public ProgressCheck
{
public:
    explicit CallbackProgressCheck(void (*cb)(int, int))
        : m_cb(cb)
    {
    }
    void Progress(int voxel, int nVoxels)
    {
        m_cb(voxel, nVoxels);
    }

private:
    void (*m_cb)(int, int);
};

/** float32 voxel data as the C ABI hands it over, sized without being written (see NEWMAT::DefaultInitAllocator) */
typedef std::vector<float, NEWMAT::DefaultInitAllocator<float> > FabberF32Values;

class FabberRunData : public Loggable
{
public:
    static void GetOptions(std::vector<OptionSpec> &opts);

    /** compat_options: the CLI's backwards-compatible default outputs (save-mean, save-std,
     *  save-zstat, save-noise-mean, save-noise-std, save-free-energy, save-mvn) */
    FabberRunData(bool compat_options = true);
    virtual ~FabberRunData();

    void Run(ProgressCheck *check = 0);

    // ---- options ----
    void Parse(int argc, char **argv);
    void ParseParamFile(const std::string &file);
    void Set(const std::string &key, const std::string &value);
    void Set(const std::string &key, double value);
    void Unset(const std::string &key);
    void SetBool(const std::string &key, bool value = true);
    bool HaveKey(const std::string &key);
    std::string GetString(const std::string &key);
    std::string GetStringDefault(const std::string &key, const std::string &def) const;
    std::vector<std::string> GetStringList(const std::string &prefix);
    bool GetBool(const std::string &key);
    int GetInt(const std::string &key, int min = INT_MIN, int max = INT_MAX);
    int GetIntDefault(const std::string &key, int def, int min = INT_MIN, int max = INT_MAX);
    std::vector<int> GetIntList(const std::string &prefix, int min = INT_MIN, int max = INT_MAX);
    double GetDouble(const std::string &key, double min = -DBL_MAX, double max = DBL_MAX);
    double GetDoubleDefault(const std::string &key, double def, double min = -DBL_MAX, double max = DBL_MAX);
    std::vector<double> GetDoubleList(const std::string &prefix, double min = -DBL_MAX, double max = DBL_MAX);
    std::string GetOutputDir();

    // ---- voxel data ----
    virtual void SaveVoxelData(const std::string &filename, NEWMAT::Matrix &data, VoxelDataType data_type = VDT_SCALAR);
    /** SaveVoxelData for a matrix the caller is done with: run data that keeps its results in memory may take the
     * storage instead of copying it (`data` is left in an unspecified state). Default: SaveVoxelData. */
    virtual void SaveVoxelDataMove(const std::string &filename, NEWMAT::Matrix &data, VoxelDataType data_type = VDT_SCALAR)
    {
        SaveVoxelData(filename, data, data_type);
    }
    const NEWMAT::Matrix &GetVoxelCoords();
    const NEWMAT::Matrix &GetVoxelData(const std::string &key);
    virtual const NEWMAT::Matrix &LoadVoxelData(const std::string &key);
    int GetVoxelDataSize(const std::string &key);
    const NEWMAT::Matrix &GetMainVoxelData();
    const NEWMAT::Matrix &GetVoxelSuppData();
    virtual void GetExtent(std::vector<int> &extent, std::vector<float> &dims);
    void SetExtent(int nx, int ny, int nz, float sx = 1.0, float sy = 1.0, float sz = 1.0);
    virtual void ClearVoxelData(std::string key = "");
    virtual void SetVoxelData(std::string key, const NEWMAT::Matrix &data);
    void SetVoxelCoords(const NEWMAT::Matrix &coords);
    /**
     * Voxel data as the C ABI hands it over: float32, [rows][voxels]. It is kept as it is - the engine reads float32
     * series directly - and becomes a (double) Matrix only if somebody asks for one through GetVoxelData /
     * LoadVoxelData. (The reference converts every volume to a NEWMAT matrix on arrival, rundata_array.cc:100-133:
     * for a million voxels x 100 timepoints that is 800 MB of freshly faulted memory before anything is computed.)
     */
    void SetVoxelDataF32(std::string key, int rows, FabberF32Values &&values);
    /** The main series (key "data") as float32 [rows][cols] if that is how it is held, else NULL */
    const float *GetMainVoxelDataF32(int &rows, int &cols);

    void Progress(int voxel, int nVoxels)
    {
        if (m_progress)
            m_progress->Progress(voxel, nVoxels);
    }
    void LogParams();
    friend std::ostream &operator<<(std::ostream &out, const FabberRunData &opts);

#ifdef DEPRECATED
    std::string Read(const std::string &key);
    std::string Read(const std::string &key, const std::string &msg);
    std::string ReadWithDefault(const std::string &key, const std::string &def);
    bool ReadBool(const std::string &key);
    void ParseOldStyleParamFile(const std::string &filename);
#endif

protected:
    void init(bool compat_options);
    void AddKeyEqualsValue(const std::string &key, bool trim_comments = false);
    void CheckAllOptionsUsed() const;
    const NEWMAT::Matrix &GetMainVoxelDataMultiple();
    void CheckSize(std::string key, const NEWMAT::Matrix &mat);

    std::map<std::string, NEWMAT::Matrix> m_voxel_data;
    struct F32Image
    {
        int rows;
        FabberF32Values values; // [rows][voxels]
    };
    std::map<std::string, F32Image> m_voxel_data_f32; // what has not been asked for as a Matrix (yet)
    std::vector<int> m_extent;
    std::vector<float> m_dims;
    ProgressCheck *m_progress;
    NEWMAT::Matrix m_empty;
    NEWMAT::Matrix m_mainDataMultiple;
    std::map<std::string, std::string> m_params;
    mutable std::set<std::string> m_used_params;
    std::string m_outdir;
    EasyLog m_default_log;
};

// ---- exceptions (also used as control flow, e.g. DataNotFound = "option absent") ----
class FabberError : public std::runtime_error
{
public:
    explicit FabberError(std::string msg)
        : std::runtime_error(msg)
        , m_msg(msg)
    {
    }
    virtual ~FabberError() throw(){};
    virtual const char *what() const throw()
    {
        return m_msg.c_str();
    }
    std::string m_msg;
};
class FabberInternalError : public FabberError
{
public:
    explicit FabberInternalError(std::string msg)
        : FabberError(msg)
    {
        m_msg = "Internal error in Fabber: " + msg;
    }
};
class FabberRunDataError : public FabberError
{
public:
    explicit FabberRunDataError(std::string msg)
        : FabberError(msg)
    {
    }
};
class InvalidOptionValue : public FabberRunDataError
{
public:
    InvalidOptionValue(std::string key, std::string value, std::string reason = "")
        : FabberRunDataError(key)
    {
        m_msg = "Invalid value given for option: " + key + "=" + value + " (" + reason + ")";
    }
};
class MandatoryOptionMissing : public FabberRunDataError
{
public:
    explicit MandatoryOptionMissing(std::string key)
        : FabberRunDataError(key)
    {
        m_msg = "No value given for mandatory option: " + key;
    }
};
class DataNotFound : public FabberRunDataError
{
public:
    DataNotFound(std::string key, std::string reason = "")
        : FabberRunDataError(key)
    {
        m_msg = "Voxel data not found: " + key + " (" + reason + ")";
    }
};

/** String -> T; the whole string must be consumed */
template <typename T> inline T convertTo(const std::string &s, const std::string &key = "")
{
    T x;
    std::istringstream i(s);
    char c;
    if (!(i >> x) || (i.get(c)))
        throw InvalidOptionValue(key, s, "Failed to convert to required type");
    return x;
}

#ifdef DEPRECATED
typedef class FabberRunData ArgsType;
typedef class FabberRunData EasyOptions;
typedef class FabberError Invalid_option;
#endif
