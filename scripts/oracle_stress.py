"""Randomised oracle-vs-GPU campaign: N random scenes (tests/stress_scenes.py) through tests/test_gpu_parity.py::_check
(forward bit-exact in both binning modes, instance lists, every gradient).  The same scenes run inside the test suite as
tests/test_gpu_stress.py (OLSR_STRESS_SCENES of them, a dozen by default); this script is the long form.

    python scripts/oracle_stress.py [N=200] [seed0=0] [vary|big]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from online_lang_splatting_amd import _C as hip, _abi
from oracle import oracle_C as oracle
import test_gpu_parity as T
from stress_scenes import random_scene

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
generation = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] in ("vary", "big") else "base"
t0 = time.time()
fails = 0
notes = 0
chain_ok = 0
rounding_only = 0
only = [int(x) for x in os.environ.get("OLSR_STRESS_ONLY", "").split(",") if x]   # (re-run single scenes of a campaign)
for k in (only or range(N)):
    sc, tile, mode, kw, desc = random_scene(k, seed0, generation)
    try:
        T._check(hip, oracle, sc, seed=k, tile=tile, mode=mode, **kw)
    except AssertionError as e:
        # the max-norm bound (1e-4 of the tensor's largest magnitude) is what the test suite asserts on its fixed scenes;
        # a breach here is re-judged by the north-star criterion per element (>= 99.99 % within 1e-4, worst <= 2e-2)
        note = f"{desc}: {str(e)[:200]}"
        try:
            T._check(hip, oracle, sc, seed=k, tile=tile, mode=mode, elementwise=True, worst_bound=2e-2, **kw)
            notes += 1
            print("NOTE (max-norm only) " + note, flush=True)
        except AssertionError as e2:
            # a few elements of the per-Gaussian chain outside the band: is it the chain's arithmetic, or its sensitivity
            # to the summation order of its inputs?  Composite-level gradients per element + the reference's chain
            # replayed on the product's own composite-level gradients (identical inputs on both sides).
            try:
                T._check(hip, oracle, sc, seed=k, tile=tile, mode=mode, elementwise=True, worst_bound=2e-2,
                         grad_keys=T.COMPOSITE_KEYS, chain=True, **kw)
                chain_ok += 1
                print("CHAIN-SENSITIVITY " + note + " | per element: " + str(e2)[:160], flush=True)
            except AssertionError as e3:
                # Round 6: an element of a composite-level tensor outside the 1e-4 band.  Is it rounding?  The reference's own
                # association on the GPU (olsr_debug_backward_ordered) must equal the oracle, and the fast kernel must lie within
                # K x 2^-24 x the element's condition of it (tests/parity_common.py: assert_rounding_only, K <= 64): then the
                # element is a sum whose terms cancel to 1e-4 of their magnitudes, and no association could do better.
                try:
                    import torch
                    from parity_common import assert_ordered_equals_oracle, assert_rounding_only, ordered_backward, run_backend
                    fo, go = run_backend(oracle, sc, None, k, tile, mode, **kw)
                    fr, gr = run_backend(hip, sc, torch.device("cuda:0"), k, tile, mode, binning=_abi.BINNING_RECT, **kw)
                    gord = ordered_backward(hip, sc, fr, k, tile, mode, **kw)
                    assert_ordered_equals_oracle(go, gord)
                    Ks = assert_rounding_only(gr, gord, ordered_backward(hip, sc, fr, k, tile, mode, condition=True, **kw), k_bound=64.0)
                    oracle.release(fo["geom"])
                    rounding_only += 1
                    print(f"ROUNDING-ONLY (K <= {max(Ks.values()):.1f}) " + note, flush=True)
                except AssertionError as e4:
                    fails += 1
                    print("FAIL " + note + " | per element: " + str(e2)[:160] + " | chain: " + str(e3)[:200] + " | rounding: " + str(e4)[:200], flush=True)
    finally:
        hip.TILE, hip.BWD_MODE, hip.BINNING = 15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE
        oracle.TILE, oracle.BWD_MODE = 15, 0
    if (k + 1) % 25 == 0:
        print(f"{k + 1}/{N} scenes, {fails} failures, {notes} notes, {chain_ok} chain-sensitivity, {time.time() - t0:.0f} s", flush=True)
print(f"done: {N} scenes, {fails} failures, {rounding_only} scenes with a composite-level element outside 1e-4 that is rounding only "
      f"(ordered kernel == oracle, fast kernel within K <= 64 of the element's condition), {notes} max-norm notes, {chain_ok} scenes where only the per-Gaussian chain's "
      f"input sensitivity shows (composite-level gradients and the chain on identical inputs both within the criterion)")
sys.exit(1 if fails else 0)
