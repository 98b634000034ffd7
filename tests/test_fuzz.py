"""GPU: a few fuzz cases of the HIP path against the CPU oracle (tests/fuzz_vs_oracle.py), NaN-poisoned scratch."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_batches_and_variants_match_the_oracle():
    from fuzz_vs_oracle import run_cases

    assert run_cases(seed=11, cases=5, verbose=False) == []
