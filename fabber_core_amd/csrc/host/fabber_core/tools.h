/* tools.h - small numerical / file helpers (reference: tools.h, tools.cc:27-98) */
#pragma once

#include "armawrap/newmat.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace fabber
{
/** Read a VEST (/NumWaves /NumPoints /Matrix) or plain ASCII ('#' comments) matrix file */
NEWMAT::Matrix read_matrix_file(const std::string &filename);
}
double gammaln(double x);
double digamma_fp64(double x);

/** body(i) for i = 0 .. n-1 on a few host threads (contiguous ranges of i per thread). For the passes over whole
 * volumes at the C ABI - float <-> double conversion, masking - which are memory-bound and, on first touch of a
 * large buffer, page-fault-bound: both scale with threads. */
template <class Body>
void fabber_parallel_for(int n, Body body, int max_threads = 16)
{
    int nt = (int)std::min<unsigned>((unsigned)max_threads, std::max(1u, std::thread::hardware_concurrency()));
    nt = std::max(1, std::min(nt, n));
    if (nt == 1)
    {
        for (int i = 0; i < n; i++)
            body(i);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
        pool.emplace_back([=]() {
            const int i0 = (int)((long long)n * t / nt), i1 = (int)((long long)n * (t + 1) / nt);
            for (int i = i0; i < i1; i++)
                body(i);
        });
    for (auto &th : pool)
        th.join();
}

/** FVB_HOST_TIMING=1: stage times of the host side of a run on stderr */
struct FabberStageTimer
{
    bool on;
    const char *who;
    std::chrono::steady_clock::time_point last;
    explicit FabberStageTimer(const char *who_)
        : on(getenv("FVB_HOST_TIMING") != NULL)
        , who(who_)
        , last(std::chrono::steady_clock::now())
    {
    }
    void lap(const char *what)
    {
        if (!on)
            return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fabber host] %s: %s %.1f ms\n", who, what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};
