"""The reference's analytic known-answer tests, applied to the CPU oracle (pins the oracle).

Sources: test/test_inference.cc:108-238,353-429,485-561; test/test_vb.cc:118-232,305-409.
"""
import pytest

import cases
import oracle


@pytest.mark.parametrize("case", cases.ALL_CASES, ids=lambda c: c.__name__)
def test_known_answer(case):
    case(oracle.run)
