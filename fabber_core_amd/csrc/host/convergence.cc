// convergence.cc - named convergence detectors on top of the shared state machine (vb_math.h)
#include "convergence.h"

#include "../vb_math.h"

using namespace std;

struct ConvergenceDetector::State
{
    fvb::ConvState s;
    int max_iterations;
    int max_trials;
    double min_fchange;
};

ConvergenceDetector::ConvergenceDetector(int type)
    : m_type(type)
    , m_state(new State())
{
    m_state->max_iterations = 10;
    m_state->max_trials = 10;
    m_state->min_fchange = 0.01;
    fvb::conv_init(m_state->s, m_type, 10, 10, 0.01);
    fvb::conv_reset(m_state->s);
}

ConvergenceDetector *ConvergenceDetector::NewFromName(const string &name)
{
    ConvergenceDetector *c = ConvergenceDetectorFactory::GetInstance()->Create(name);
    if (!c)
        throw InvalidOptionValue("convergence", name, "Unrecognized convergence detector");
    return c;
}

void ConvergenceDetector::Initialize(FabberRunData &params)
{
    m_log = params.GetLogger();
    m_state->max_iterations = convertTo<int>(params.GetStringDefault("max-iterations", "10"));
    if (m_state->max_iterations <= 0)
        throw InvalidOptionValue("max-iterations", stringify(m_state->max_iterations), "Must be positive");
    if (m_type == FVB_CONV_LM)
    {
        m_state->min_fchange = convertTo<double>(params.GetStringDefault("max-fchange", "0.01"));
        if (m_state->min_fchange <= 0)
            throw InvalidOptionValue("max-fchange", stringify(m_state->min_fchange), "Must be positive");
    }
    else if (m_type != FVB_CONV_MAXITS)
    {
        m_state->min_fchange = convertTo<double>(params.GetStringDefault("min-fchange", "0.01"));
        if (m_state->min_fchange <= 0)
            throw InvalidOptionValue("min-fchange", stringify(m_state->min_fchange), "Must be positive");
    }
    if (m_type == FVB_CONV_TRIALMODE)
    {
        m_state->max_trials = convertTo<int>(params.GetStringDefault("max-trials", "10"));
        if (m_state->max_trials <= 0)
            throw InvalidOptionValue("max-trials", stringify(m_state->max_trials), "Must be positive");
    }
    fvb::conv_init(m_state->s, m_type, m_state->max_iterations, m_state->max_trials, m_state->min_fchange);
    fvb::conv_reset(m_state->s);
}

bool ConvergenceDetector::Test(double F)
{
    return fvb::conv_test(m_state->s, F);
}
void ConvergenceDetector::Reset(double F)
{
    fvb::conv_reset(m_state->s);
    m_state->s.prev_f = F;
}
bool ConvergenceDetector::UseF() const
{
    return m_type != FVB_CONV_MAXITS;
}
bool ConvergenceDetector::NeedSave()
{
    return fvb::conv_need_save(m_state->s);
}
bool ConvergenceDetector::NeedRevert()
{
    return fvb::conv_need_revert(m_state->s);
}
float ConvergenceDetector::LMalpha()
{
    return (float)fvb::conv_lm_alpha(m_state->s);
}
int ConvergenceDetector::MaxIterations() const
{
    return m_state->max_iterations;
}
int ConvergenceDetector::MaxTrials() const
{
    return m_state->max_trials;
}
double ConvergenceDetector::MinFChange() const
{
    return m_state->min_fchange;
}
void ConvergenceDetector::Dump(ostream &out, const string &indent) const
{
    out << indent << "Iteration " << m_state->s.its << " of at most " << m_state->s.max_its << endl;
    out << indent << "Previous Free Energy == " << m_state->s.prev_f << endl;
}

namespace
{
struct Det : ConvergenceDetector
{
    explicit Det(int type)
        : ConvergenceDetector(type)
    {
    }
};
}
ConvergenceDetector *ConvergenceDetector::NewMaxIts()
{
    return new Det(FVB_CONV_MAXITS);
}
ConvergenceDetector *ConvergenceDetector::NewFchange()
{
    return new Det(FVB_CONV_FCHANGE);
}
ConvergenceDetector *ConvergenceDetector::NewFreduce()
{
    return new Det(FVB_CONV_FREDUCE);
}
ConvergenceDetector *ConvergenceDetector::NewTrialMode()
{
    return new Det(FVB_CONV_TRIALMODE);
}
ConvergenceDetector *ConvergenceDetector::NewLM()
{
    return new Det(FVB_CONV_LM);
}
