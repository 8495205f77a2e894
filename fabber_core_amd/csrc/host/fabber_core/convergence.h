/* convergence.h - convergence detectors by name (reference: convergence.h, setup.cc:50-58).
 * One class drives the same POD state machine the HIP kernels run (../../vb_math.h). */
#pragma once

#include "factories.h"
#include "rundata.h"

#include <ostream>
#include <string>

class ConvergenceDetector : public Loggable
{
public:
    static ConvergenceDetector *NewFromName(const std::string &name);
    virtual ~ConvergenceDetector()
    {
    }
    virtual void Initialize(FabberRunData &params);
    virtual bool Test(double F);
    virtual void Reset(double F = -99e99);
    virtual bool UseF() const;
    virtual bool NeedSave();
    virtual bool NeedRevert();
    virtual float LMalpha();
    std::string GetReason()
    {
        return m_reason;
    }
    virtual void Dump(std::ostream &out, const std::string &indent = "") const;
    /** engine code: enum fvb_convergence */
    int DeviceCode() const
    {
        return m_type;
    }
    int MaxIterations() const;
    int MaxTrials() const;
    double MinFChange() const;

    // one constructor function per registered name
    static ConvergenceDetector *NewMaxIts();
    static ConvergenceDetector *NewFchange();
    static ConvergenceDetector *NewFreduce();
    static ConvergenceDetector *NewTrialMode();
    static ConvergenceDetector *NewLM();

protected:
    explicit ConvergenceDetector(int type);
    int m_type;
    std::string m_reason;
    struct State;
    State *m_state;
};

typedef SingletonFactory<ConvergenceDetector> ConvergenceDetectorFactory;
