/* easylog.h - log stream + once-only warnings used by every fabber component.
 * API-compatible with the reference's EasyLog / Loggable (easylog.h:21-118): LOG, LOG_ERR,
 * WARN_ONCE, WARN_ALWAYS macros, stringify<>(). Host-side only. */
#pragma once

#include <assert.h>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#define LOG ((m_log == 0) ? std::cerr : m_log->LogStream())
#define LOG_ERR(x) ((void)((LOG << x)))
#define WARN_ONCE(x)                                                                                         \
    if (m_log)                                                                                               \
    m_log->WarnOnce(x)
#define WARN_ALWAYS(x)                                                                                       \
    if (m_log)                                                                                               \
    m_log->WarnAlways(x)

class EasyLog
{
public:
    EasyLog();
    ~EasyLog();
    /** Log to <outDir>/logfile */
    void StartLog(const std::string &outDir);
    /** Log to an existing stream */
    void StartLog(std::ostream &s);
    const std::string &GetOutputDirectory();
    void StopLog(bool gzip = false);
    bool LogStarted();
    /** Stream to write to; before StartLog everything is buffered and replayed on start */
    std::ostream &LogStream();
    void WarnOnce(const std::string &text);
    void WarnAlways(const std::string &text);
    /** Repeat every warning seen so far with its count */
    void ReissueWarnings();

private:
    std::ostream *m_stream;
    bool m_owns_stream;
    std::stringstream m_templog;
    std::string m_outdir;
    std::map<std::string, int> m_warncount;
};

class Loggable
{
public:
    explicit Loggable(EasyLog *log = 0)
        : m_log(log)
        , m_debug(false)
    {
    }
    EasyLog *GetLogger() const
    {
        return m_log;
    }
    void SetLogger(EasyLog *log)
    {
        m_log = log;
    }

protected:
    EasyLog *m_log;
    bool m_debug;
};

template <typename type> inline std::string stringify(type from)
{
    std::ostringstream s;
    if (!(s << from))
        throw std::logic_error("Stringify failed");
    return s.str();
}

inline std::ostream &operator<<(std::ostream &out, std::vector<int> x)
{
    out << "[ ";
    for (unsigned i = 0; i < x.size(); i++)
        out << x[i] << " ";
    out << "]";
    return out;
}
