"""A slice of the randomised oracle-vs-GPU campaign inside the suite: OLSR_STRESS_SCENES random
scenes per generation (default 6; the campaign of scripts/oracle_stress.py ran 9 300) through the full `_check` — forward
bit-identical in both binning modes, instance lists, every gradient.  A breach of the max-norm bound is re-judged by
the north-star criterion per element, exactly as the campaign did (screen-filling splats amplify summation-order noise
in the inverse of the 2D covariance)."""
import os

import pytest

from stress_scenes import random_room_scene, random_scene
from test_gpu_parity import COMPOSITE_KEYS, _check

pytestmark = pytest.mark.gpu
N = int(os.environ.get("OLSR_STRESS_SCENES", "6"))
SEED0 = int(os.environ.get("OLSR_STRESS_SEED0", "20000"))


@pytest.mark.parametrize("generation", ["base", "vary", "room"])
@pytest.mark.parametrize("k", range(N))
def test_random_scene_against_the_oracle(hip, oracle, generation, k):
    # ("room", round 5: surface-structured maps built by the reference's recipe, fresh or perturbed like an optimised one)
    sc, tile, mode, kw, desc = random_room_scene(k, SEED0) if generation == "room" else random_scene(k, SEED0, generation)
    try:
        _check(hip, oracle, sc, seed=k, tile=tile, mode=mode, **kw)
    except AssertionError as e:
        head = str(e)[:120]
        if "not bit-identical" in head or "forward" in head or ":chain:" in head or any(c in head for c in COMPOSITE_KEYS):
            raise  # the forward, everything the composite kernel produces and the chain on identical inputs are never re-judged
        print("max-norm breach behind the per-Gaussian chain, re-judged per element:", desc, str(e)[:160])
        rejudged = "per element"
        try:
            _check(hip, oracle, sc, seed=k, tile=tile, mode=mode, elementwise=True, worst_bound=2e-2, **kw)
        except AssertionError as e2:
            rejudged = "chain on identical inputs"
            # ~2 % of the campaign's scenes (screen-filling splats, random precomputed covariances): a few elements per
            # 10^4 of the gradients behind the inverse of the 2D covariance leave the band — three terms of order 1e8 cancel
            # there, so the chain amplifies the summation-order noise of its INPUTS (the reference's own float atomics
            # have the same noise from run to run).  What must still hold to the element:
            # everything the composite produces ...
            if "forward" in str(e2) or "bit-identical" in str(e2):
                raise
            print("per-element breach in the covariance chain:", desc, str(e2)[:200])
            # ... and the per-Gaussian chain itself once both sides start from the same composite-level gradients (chain=True,
            # the default of _check: the oracle replays the reference's chain on the product's dL_dconic / dL_dmean2D).
            _check(hip, oracle, sc, seed=k, tile=tile, mode=mode, elementwise=True, worst_bound=2e-2,
                   grad_keys=COMPOSITE_KEYS, chain=True, **kw)
        # a re-judged scene is reported, not passed silently: it shows as `x` with its reason
        pytest.xfail(f"{desc}: end-to-end max-norm breach in the covariance chain, accepted by the weaker criterion "
                     f"'{rejudged}' ({str(e)[:120]})")
