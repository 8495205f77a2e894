/* setup.h - registration of the built-in components (reference: setup.h, setup.cc:26-73) */
#pragma once

class FabberSetup
{
public:
    static void SetupDefaults();
    static void SetupDefaultInferenceTechniques();
    static void SetupDefaultNoiseModels();
    static void SetupDefaultFwdModels();
    static void SetupDefaultConvergenceDetectors();
    static void Destroy();
};
