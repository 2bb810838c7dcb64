"""The oracle against its own frozen outputs (tests/golden/oracle_freeze.json, written by tests/golden/make_oracle_freeze.py): the
reference pins almost nothing numerically on this path, so what CAN be pinned is that the restatement does
not drift -- any edit that changes a distance bit, a tie-break or an id order shows up here and has to be argued from the reference."""
import importlib.util
import json
import os


def test_the_oracle_still_computes_what_was_frozen(oracle):
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_freeze", os.path.join(here, "make_oracle_freeze.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(here, "oracle_freeze.json")) as f:
        want = json.load(f)
    got = mod.compute()
    assert sorted(got) == sorted(want)
    changed = [k for k in want if got[k] != want[k]]
    assert not changed, f"the oracle's results changed for: {changed}"
