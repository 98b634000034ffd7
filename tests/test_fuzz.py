"""GPU: fuzz cases of the HIP path against the CPU oracle (tests/fuzz_vs_oracle.py), NaN-poisoned scratch: random ragged
batches (a quarter of them 32-64 scenes of up to 32 pedestrians), generator / sample counts, loss masks and variants."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,cases", [(11, 7), (29, 7)])
def test_random_batches_and_variants_match_the_oracle(seed, cases):
    from fuzz_vs_oracle import run_cases

    assert run_cases(seed=seed, cases=cases, verbose=False) == []
