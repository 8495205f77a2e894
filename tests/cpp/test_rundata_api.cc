// test_rundata_api.cc - the C++ class API model libraries and embedding programs use
// (FabberRunData options / voxel data, exceptions), checked with the cases of the reference's
// test/test_rundata.cc:44-505. Built and run by tests/test_cpp_api.py; no GPU involved.
#include "fabber_core/easylog.h"
#include "fabber_core/rundata.h"
#include "fabber_core/setup.h"

#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

using namespace std;
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
#define CHECK_THROWS(expr, ExcType)                                                                          \
    do                                                                                                       \
    {                                                                                                        \
        bool thrown_ = false;                                                                                \
        try                                                                                                  \
        {                                                                                                    \
            expr;                                                                                            \
        }                                                                                                    \
        catch (ExcType &)                                                                                    \
        {                                                                                                    \
            thrown_ = true;                                                                                  \
        }                                                                                                    \
        catch (...)                                                                                          \
        {                                                                                                    \
        }                                                                                                    \
        if (!thrown_)                                                                                        \
        {                                                                                                    \
            printf("FAILED %s:%d: %s did not throw %s\n", __FILE__, __LINE__, #expr, #ExcType);              \
            g_failures++;                                                                                    \
        }                                                                                                    \
    } while (0)

static const int NTIMES = 10, VSIZE = 5, NVOX = VSIZE * VSIZE * VSIZE;
static const float VAL = 7.32f;

static void cube(Matrix &coords, Matrix *d1, Matrix *d2, Matrix *d3)
{
    coords.ReSize(3, NVOX);
    for (Matrix *d : { d1, d2, d3 })
        if (d)
            d->ReSize(NTIMES, NVOX);
    int v = 1;
    for (int z = 0; z < VSIZE; z++)
        for (int y = 0; y < VSIZE; y++)
            for (int x = 0; x < VSIZE; x++, v++)
            {
                coords(1, v) = x;
                coords(2, v) = y;
                coords(3, v) = z;
                for (int n = 1; n <= NTIMES; n++)
                {
                    if (d1)
                        (*d1)(n, v) = VAL;
                    if (d2)
                        (*d2)(n, v) = VAL * 2;
                    if (d3)
                        (*d3)(n, v) = VAL * 3;
                }
            }
}

static bool feq(double a, double b)
{
    return fabs(a - b) <= 4e-7 * fabs(b);
}

static void multi_data()
{
    Matrix coords, d1, d2, d3;
    cube(coords, &d1, &d2, &d3);
    for (const char *order : { "concatenate", "interleave" })
    {
        EasyLog log;
        FabberRunData rundata;
        rundata.SetLogger(&log);
        rundata.SetVoxelCoords(coords);
        rundata.SetVoxelData("data1", d1);
        rundata.SetVoxelData("data2", d2);
        rundata.SetVoxelData("data3", d3);
        rundata.Set("data-order", order);
        Matrix data = rundata.GetMainVoxelData();
        CHECK(data.Nrows() == NTIMES * 3 && data.Ncols() == NVOX);
        bool ok = true;
        for (int i = 1; i <= NVOX && ok; i++)
            for (int t = 0; t < NTIMES * 3 && ok; t++)
            {
                const int which = string(order) == "concatenate" ? t / NTIMES : t % 3;
                ok = feq(data(t + 1, i), VAL * (which + 1));
            }
        CHECK(ok);
    }
    FabberRunData rundata;
    rundata.SetVoxelCoords(coords);
    rundata.SetVoxelData("data1", d1);
    rundata.SetVoxelData("data2", d2);
    rundata.SetVoxelData("data3", d3);
    rundata.Set("data-order", "singlefile");
    CHECK_THROWS(Matrix data = rundata.GetMainVoxelData(), InvalidOptionValue);
}

static void options_files()
{
    const string fname = "test_config_tmp";
    {
        ofstream os(fname.c_str());
        os << "noise=white" << endl << "model=poly" << endl << "method=vb" << endl << "bool-option" << endl << "#comment, ignored" << endl;
    }
    FabberRunData rundata;
    rundata.ParseParamFile(fname);
    CHECK(rundata.GetString("noise") == "white");
    CHECK(rundata.GetString("model") == "poly");
    CHECK(rundata.GetString("method") == "vb");
    CHECK(rundata.GetBool("bool-option"));
    {
        ofstream os(fname.c_str());
        os << "model=poly" << endl << "degree=0 # Keep things simple" << endl;
    }
    FabberRunData r2;
    r2.ParseParamFile(fname);
    CHECK(r2.GetString("model") == "poly");
    CHECK(r2.GetInt("degree") == 0);
    remove(fname.c_str());
}

static void option_values()
{
    FabberRunData rundata;
    rundata.Set("wibble", "wobble");
    rundata.SetBool("bobble");
    CHECK(rundata.GetStringDefault("wibble", "squabble") == "wobble");
    rundata.Unset("wibble");
    CHECK(rundata.GetStringDefault("wibble", "squabble") == "squabble");
    CHECK(rundata.GetBool("bobble"));
    rundata.Unset("bobble");
    CHECK(!rundata.GetBool("bobble"));

    rundata.Set("i", "7");
    CHECK(rundata.GetInt("i") == 7 && rundata.GetInt("i", 7, 8) == 7 && rundata.GetInt("i", 6, 7) == 7 && rundata.GetInt("i", 7, 7) == 7);
    CHECK_THROWS(rundata.GetInt("i", 10), InvalidOptionValue);
    CHECK_THROWS(rundata.GetInt("i", 0, 3), InvalidOptionValue);
    rundata.Set("d", "7.5");
    CHECK(rundata.GetDouble("d") == 7.5 && rundata.GetDouble("d", 7, 8) == 7.5);
    CHECK_THROWS(rundata.GetDouble("d", 10.1), InvalidOptionValue);
    CHECK_THROWS(rundata.GetDouble("d", 0.2, 3.3), InvalidOptionValue);
    rundata.Set("bad", "ABC");
    CHECK_THROWS(rundata.GetInt("bad"), InvalidOptionValue);
    CHECK_THROWS(rundata.GetDouble("bad"), InvalidOptionValue);

    rundata.Set("il1", "7");
    rundata.Set("il2", "8");
    vector<int> il = rundata.GetIntList("il");
    CHECK(il.size() == 2 && il[0] == 7 && il[1] == 8);
    rundata.Set("dl1", "7.5");
    rundata.Set("dl2", "8.5");
    vector<double> dl = rundata.GetDoubleList("dl");
    CHECK(dl.size() == 2 && dl[0] == 7.5 && dl[1] == 8.5);
    rundata.Set("bl1", "2");
    rundata.Set("bl2", "ABC");
    CHECK_THROWS(rundata.GetIntList("bl"), InvalidOptionValue);
    CHECK_THROWS(rundata.GetDoubleList("bl"), InvalidOptionValue);
    rundata.Set("ml1", "14");
    rundata.Set("ml2", "7");
    CHECK_THROWS(rundata.GetIntList("ml", 10), InvalidOptionValue);
    CHECK_THROWS(rundata.GetDoubleList("ml", 10), InvalidOptionValue);
    rundata.Set("xl1", "1");
    rundata.Set("xl2", "2");
    rundata.Set("xl3", "7");
    CHECK_THROWS(rundata.GetIntList("xl", 0, 3), InvalidOptionValue);
    CHECK_THROWS(rundata.GetDoubleList("xl", 0, 3), InvalidOptionValue);
}

static void voxel_data()
{
    Matrix coords, d1, d2, d3;
    cube(coords, &d1, &d2, &d3);
    {
        // a data set called "data" and the option data=data must not recurse (:422-460)
        FabberRunData rundata;
        rundata.SetVoxelCoords(coords);
        rundata.SetVoxelData("data", d1);
        rundata.Set("data", "data");
        Matrix data = rundata.GetMainVoxelData();
        CHECK(data.Ncols() == NVOX && data.Nrows() == NTIMES);
    }
    FabberRunData rundata;
    rundata.SetVoxelCoords(coords);
    rundata.SetVoxelData("data1", d1);
    rundata.SetVoxelData("data2", d2);
    rundata.SetVoxelData("data3", d3);
    rundata.ClearVoxelData("data1");
    rundata.GetVoxelCoords();
    CHECK_THROWS(rundata.GetVoxelData("data1"), DataNotFound);
    CHECK(rundata.GetVoxelData("data2").Ncols() == NVOX);
    CHECK(rundata.GetVoxelData("data3").Ncols() == NVOX);
    rundata.ClearVoxelData();
    CHECK_THROWS(rundata.GetVoxelCoords(), DataNotFound);
    CHECK_THROWS(rundata.GetVoxelData("data2"), DataNotFound);
    CHECK_THROWS(rundata.GetVoxelData("data3"), DataNotFound);
}

int main()
{
    FabberSetup::SetupDefaults();
    multi_data();
    options_files();
    option_values();
    voxel_data();
    FabberSetup::Destroy();
    printf(g_failures ? "%d check(s) failed\n" : "all checks passed\n", g_failures);
    return g_failures ? 1 : 0;
}
