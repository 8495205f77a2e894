"""Convergence-detector state machines: the reference's test/test_convergence.cc:35-305, run
against every implementation of the state machine (the CPU oracle's classes and the product's
host-compiled twin of the device state machine).
"""
import numpy as np
import pytest

import oracle

MAXITERS = 37
MAXTRIALS = 3
FCHANGE = 0.0001
F0 = 12.1


def _oracle_impl(conv, F):
    done, _, _, _ = oracle.convergence_trace(conv, F, max_iterations=MAXITERS, max_trials=MAXTRIALS,
                                             min_fchange=FCHANGE, stop_at_done=False)
    return list(done)


def _product_impl(conv, F):
    from fabber_core_amd import hiplib
    done, _, _, _ = hiplib.convergence_trace(conv, F, max_iterations=MAXITERS, max_trials=MAXTRIALS,
                                             min_fchange=FCHANGE, stop_at_done=False)
    return list(done)


IMPLS = {"oracle": _oracle_impl, "product": _product_impl}


@pytest.fixture(params=sorted(IMPLS))
def impl(request):
    if request.param == "product":
        from fabber_core_amd import hiplib
        if not hiplib.available():
            pytest.skip("product library not built")
    return IMPLS[request.param]


def check(impl, conv, sequence):
    """sequence: list of (F, expected Test(F) result). A detector keeps answering after it has
    said 'done' in the reference's tests, so the whole sequence is fed without stopping."""
    got = impl(conv, [f for f, _ in sequence])
    exp = [e for _, e in sequence]
    assert got == exp


def test_counting(impl):
    check(impl, "maxits", [(F0, False)] * (MAXITERS - 1) + [(F0, True)])


@pytest.mark.parametrize("conv", ["pointzeroone", "freduce"])
def test_fchange_maxiters(impl, conv):
    seq = [(F0 + 2 * i * FCHANGE, False) for i in range(MAXITERS - 1)] + [(F0 + 2 * MAXITERS * FCHANGE, True)]
    check(impl, conv, seq)


def test_fchange_change(impl):
    check(impl, "pointzeroone", [(F0, False), (F0 + 2 * FCHANGE, False), (F0, False), (F0 + 1.01 * FCHANGE, False),
                                 (F0 + 1.99 * FCHANGE, True), (F0 + 1.99 * FCHANGE, True)])
    check(impl, "pointzeroone", [(F0 + 1.99 * FCHANGE, False), (F0, False), (F0, True)])


@pytest.mark.parametrize("conv", ["freduce", "trialmode"])
def test_increase_only(impl, conv):
    check(impl, conv, [(F0, False), (F0 + 2 * FCHANGE, False), (F0 + 3.01 * FCHANGE, False),
                       (F0 + 3.99 * FCHANGE, True), (F0 + 3.99 * FCHANGE, True)])
    check(impl, conv, [(F0 + 3.99 * FCHANGE, False), (F0 + 5 * FCHANGE, False), (F0 + 5 * FCHANGE, True)])


def test_freduce_reduce(impl):
    check(impl, "freduce", [(F0, False), (F0 + 2 * FCHANGE, False), (F0 - 2 * FCHANGE, True)])
    check(impl, "freduce", [(F0 - 3 * FCHANGE, False), (F0, False), (F0 - 5 * FCHANGE, True)])


def test_trialmode_maxiters(impl):
    # one more iteration than requested (convergence.cc:145)
    seq = [(F0 + 2 * i * FCHANGE, False) for i in range(MAXITERS)] + [(F0 + 2 * MAXITERS * FCHANGE, True)]
    check(impl, "trialmode", seq)


def test_trialmode_reduce(impl):
    seq = [(F0, False), (F0 + 2 * FCHANGE, False)]
    seq += [(F0 - 2 * i * FCHANGE, False) for i in range(MAXTRIALS - 1)]
    seq += [(F0 - 2 * MAXTRIALS * FCHANGE, True)]
    check(impl, "trialmode", seq)
    # Second half of the reference's TestTrialModeConvergenceDetectorReduce
    # (test_convergence.cc:286-304). The test's comment expects the repeat of F0 + 2 FCHANGE to
    # "reset number of trials", but convergence.cc:200-243 as written compares with 'diff > 0'
    # and the repeated value gives diff == 0 exactly, so it counts as a trial and the next
    # decrease is the third trial = max-trials -> stop. The code, not the stale expectation, is
    # what the product must reproduce.
    seq = [(F0, False), (F0 + 2 * FCHANGE, False), (F0, False), (F0 + 2 * FCHANGE, False), (F0, True)]
    check(impl, "trialmode", seq)
    # With a genuine increase over the best F the trial counter does reset (convergence.cc:216-226)
    seq = [(F0, False), (F0 + 2 * FCHANGE, False), (F0, False), (F0 + 4 * FCHANGE, False)]
    seq += [(F0 - 2 * i * FCHANGE, False) for i in range(MAXTRIALS - 1)]
    seq += [(F0 - 2 * MAXTRIALS * FCHANGE, True)]
    check(impl, "trialmode", seq)


def test_lm_alpha_schedule():
    """LM detector (convergence.cc:278-378): a decrease enters LM mode with alpha 1e-6, further
    failures multiply by 10, a success at alpha > start divides by 10."""
    F = [0.0, 10.0, 5.0, 4.0, 20.0, 30.0, 30.001]
    done, save, revert, alpha = oracle.convergence_trace("lm", F, max_iterations=10, min_fchange=0.01,
                                                         stop_at_done=False)
    assert list(done) == [False, False, False, False, False, False, True]
    assert np.allclose(alpha, [0, 0, np.float32(1e-6), np.float32(1e-5), np.float32(1e-6), 1e-6, 1e-6], rtol=1e-6)
    assert list(revert) == [False, False, True, True, False, False, False]
    assert all(save)
