// test_run_class_api.cc - a C++ program using the class API end to end, as the reference's own
// tests do (test/test_inference.cc:353-429, test/test_vb.cc:49-59): FabberRunData::Run on the GPU
// engine, then the technique driven directly (Initialize / DoCalculations / SaveResults).
#include "fabber_core/fwdmodel.h"
#include "fabber_core/inference.h"
#include "fabber_core/rundata.h"
#include "fabber_core/setup.h"

#include <cmath>
#include <cstdio>
#include <memory>

using NEWMAT::Matrix;

static int g_failures = 0;
#define CHECK(cond)                                                                                          \
    do                                                                                                       \
    {                                                                                                        \
        if (!(cond))                                                                                         \
        {                                                                                                    \
            printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                                         \
            g_failures++;                                                                                    \
        }                                                                                                    \
    } while (0)

int main()
{
    const int NTIMES = 10, VSIZE = 5, NVOX = VSIZE * VSIZE * VSIZE;
    Matrix coords(3, NVOX), data(NTIMES, NVOX);
    int v = 1;
    for (int z = 0; z < VSIZE; z++)
        for (int y = 0; y < VSIZE; y++)
            for (int x = 0; x < VSIZE; x++, v++)
            {
                coords(1, v) = x;
                coords(2, v) = y;
                coords(3, v) = z;
                for (int n = 1; n <= NTIMES; n++)
                    data(n, v) = 2 + 3.0 * n * n - 4.0 * n * n * n; // cubic, test_inference.cc:353
            }
    {
        FabberRunData rundata;
        rundata.SetVoxelCoords(coords);
        rundata.SetVoxelData("data", data);
        rundata.Set("noise", "white");
        rundata.Set("model", "poly");
        rundata.Set("degree", "3");
        rundata.Set("method", "vb");
        rundata.Set("max-iterations", "50");
        rundata.SetBool("save-mean");
        rundata.Run();
        const double want[4] = { 2, 0, 3, -4 };
        for (int k = 0; k < 4; k++)
        {
            Matrix mean = rundata.GetVoxelData("mean_c" + stringify(k));
            CHECK(mean.Nrows() == 1 && mean.Ncols() == NVOX);
            bool ok = true;
            for (int i = 1; i <= NVOX; i++)
                ok = ok && std::fabs(mean(1, i) - want[k]) < 1e-3;
            CHECK(ok);
        }
    }
    {
        // the technique driven directly
        FabberSetup::SetupDefaults();
        FabberRunData rundata;
        rundata.SetVoxelCoords(coords);
        rundata.SetVoxelData("data", data);
        rundata.Set("noise", "white");
        rundata.Set("model", "poly");
        rundata.Set("degree", "3");
        rundata.Set("method", "vb"); // Vb::IsSpatial asks for it, as in the reference (inference_vb.cc:336)
        rundata.Set("max-iterations", "50");
        rundata.SetBool("save-mean");
        rundata.SetBool("save-model-fit");
        std::unique_ptr<FwdModel> model(FwdModel::NewFromName("poly"));
        model->Initialize(rundata);
        std::unique_ptr<InferenceTechnique> vb(InferenceTechnique::NewFromName("vb"));
        vb->Initialize(model.get(), rundata);
        vb->DoCalculations(rundata);
        vb->SaveResults(rundata);
        Matrix fit = rundata.GetVoxelData("modelfit");
        CHECK(fit.Nrows() == NTIMES && fit.Ncols() == NVOX);
        CHECK(std::fabs(fit(3, 7) - data(3, 7)) < 1e-2 * std::fabs(data(3, 7)));
    }
    printf(g_failures ? "%d check(s) failed\n" : "all checks passed\n", g_failures);
    return g_failures ? 1 : 0;
}
