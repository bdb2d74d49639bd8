"""What the statistical pin of the oracle can and cannot see: the committed table
tests/golden/pin_sensitivity.json (made by tests/pin_sensitivity.py: mis-restatements seeded into a copy of the oracle,
each run through the reference's own checks and through the cross-checks this repository adds) is complete, and its
cheap columns are reproduced live.  CPU only."""
import json

import numpy as np
import pytest

import pin_sensitivity as ps


@pytest.fixture(scope="module")
def table():
    return json.load(open(ps.OUT))


def test_the_mutants_are_off_in_the_oracle_proper():
    """libtgnh_oracle.so is built without -DTGO_MUTANTS: it has no switch to flip."""
    import ctypes
    from oracle import build_oracle
    lib = ctypes.CDLL(build_oracle())
    assert not hasattr(lib, "tgo_set_mutant")
    assert hasattr(ps.load_mutants(), "tgo_set_mutant")


def test_table_is_complete(table):
    assert sorted(table, key=int) == [str(k) for k in sorted(ps.MUTANTS)]
    for k, (what, modes) in ps.MUTANTS.items():
        row = table[str(k)]
        assert row["what"] == what and sorted(row["modes"]) == sorted(modes)
        for mode in modes:
            assert sorted(row["modes"][mode]) == ["bridge", "energy", "pair", "water"], (k, mode)
    # the unmutated oracle passes everything; the control (a rescale by the wrong power of the chain's factor) is seen by
    # every check that follows a trajectory -- but NOT by testWater, whose mean temperature any thermostat with the right
    # N kT reaches
    for mode in ("dualNH", "TGNH"):
        assert not any(d["caught"] for d in table["0"]["modes"][mode].values())
        assert all(table["9"]["modes"][mode][n]["caught"] for n in ("pair", "energy", "bridge"))


@pytest.mark.parametrize("name", sorted(ps.CHEAP))
def test_cheap_detectors_reproduce_the_table(table, name):
    for k, (_, modes) in ps.MUTANTS.items():
        for mode in modes:
            live = ps.CHEAP[name](k, mode)
            want = table[str(k)]["modes"][mode][name]
            assert live["caught"] == want["caught"], (name, k, mode, live, want)


def test_summary_matches_the_table(table):
    """tests/golden/pin_sensitivity_summary.json is what DESIGN.md section 6 quotes: it must follow from the table."""
    assert ps.summary(table) == json.load(open(ps.SUMMARY))
    sm = ps.summary(table)
    print("not noticed by testWater:", sm["unseen_by_testWater"])
    print("not noticed by the reference's own checks:", sm["unseen_by_reference_checks"])
    print("not noticed by anything:", sm["unseen_by_all"])
    assert [9, "dualNH"] in sm["unseen_by_testWater"] and [9, "TGNH"] in sm["unseen_by_testWater"]
    assert np.isfinite(table["0"]["modes"]["dualNH"]["water"]["deviation"])
