"""A fixed slice of tests/oracle_soak.py in the suite: the HIP path against the oracle over RANDOM configurations -- topology x mode x
precision x flag set x tile kind x chain length x every integrator parameter, the temperature groups handed over in both modes
(round 4: dualNH used them as bin indices; no test that zeroed them first could see it) -- and the same for the particle-sharded
path, two ranks on this GPU through the library's mailboxes.  The soak itself runs tens of thousands of such cases
(profiles/r04_fuzz_soak.md); these seeds are the suite's share."""
import numpy as np
import pytest

import oracle_soak

pytestmark = pytest.mark.gpu


def _run(case, seeds, nsteps):
    verdicts = {"ok": 0, "skip": 0}
    for seed in seeds:
        info = {"what": ""}
        try:
            kind = case(np.random.default_rng(seed), nsteps, info)[0]
        except oracle_soak.OracleError:
            kind = "skip"                                    # random constraint clusters the oracle's own SHAKE gives up on
        except Exception as e:
            raise AssertionError(f"seed {seed}: {info['what']}") from e
        verdicts[kind] += 1
    return verdicts


def test_random_configurations_against_the_oracle():
    v = _run(oracle_soak.one_case, range(400), 30)
    assert v["ok"] >= 350, v                                 # (the rest: refused as unsupported at create, or no verdict from the oracle)


def test_random_sharded_configurations_against_the_oracle():
    v = _run(oracle_soak.sharded_case, range(500000, 500150), 20)
    assert v["ok"] >= 130, v


def test_random_checkpoints_restore_bit_for_bit():
    v = _run(oracle_soak.checkpoint_case, range(4000000, 4000150), 15)
    assert v["ok"] >= 130, v
