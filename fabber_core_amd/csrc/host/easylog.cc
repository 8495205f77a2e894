// easylog.cc - see fabber_core/easylog.h
#include "easylog.h"

#include <fstream>

EasyLog::EasyLog()
    : m_stream(NULL)
    , m_owns_stream(false)
{
}

EasyLog::~EasyLog()
{
    StopLog();
}

void EasyLog::StartLog(const std::string &outDir)
{
    StopLog();
    m_outdir = outDir;
    std::ofstream *f = new std::ofstream((outDir + "/logfile").c_str());
    if (!f->good())
    {
        delete f;
        throw std::runtime_error("Cannot open logfile in " + outDir);
    }
    m_stream = f;
    m_owns_stream = true;
    *m_stream << m_templog.str();
    m_templog.str("");
}

void EasyLog::StartLog(std::ostream &s)
{
    StopLog();
    m_stream = &s;
    m_owns_stream = false;
    m_outdir = "";
    *m_stream << m_templog.str();
    m_templog.str("");
}

const std::string &EasyLog::GetOutputDirectory()
{
    return m_outdir;
}

void EasyLog::StopLog(bool)
{
    if (m_stream)
        m_stream->flush();
    if (m_owns_stream)
        delete m_stream;
    m_stream = NULL;
    m_owns_stream = false;
}

bool EasyLog::LogStarted()
{
    return m_stream != NULL;
}

std::ostream &EasyLog::LogStream()
{
    return m_stream ? *m_stream : m_templog;
}

void EasyLog::WarnOnce(const std::string &text)
{
    if (++m_warncount[text] == 1)
        LogStream() << "WARNING ONCE: " << text << std::endl;
}

void EasyLog::WarnAlways(const std::string &text)
{
    ++m_warncount[text];
    LogStream() << "WARNING ALWAYS: " << text << std::endl;
}

void EasyLog::ReissueWarnings()
{
    if (m_warncount.empty())
        return;
    LogStream() << "\nSummary of warnings (" << m_warncount.size() << " distinct warnings)\n";
    for (std::map<std::string, int>::iterator it = m_warncount.begin(); it != m_warncount.end(); ++it)
        LogStream() << "Issued " << (it->second == 1 ? "once: " : stringify(it->second) + " times: ") << it->first
                    << std::endl;
}
