"""The host comparison behind column predicates (cozo_amd.hnsw._compare) against the reference's operators
(cozo-core/src/data/functions.rs:298-380): op_eq / op_neq never check types, the ordering operators do."""
import numpy as np
import pytest

from cozo_amd.hnsw import _compare


def test_eq_and_neq_accept_any_pair_of_values():
    assert _compare(None, "==", 5) is False and _compare(None, "!=", 5) is True      # Null == 5 is false
    assert _compare("a", "!=", 5) is True and _compare("a", "==", "a") is True
    assert _compare(None, "==", None) is True
    assert _compare(True, "==", 1) is False                                            # Bool(true) is not Num(1)
    assert _compare([1, 2], "==", [1, 2]) and not _compare([1, 2], "==", [1, 2.0])     # Int 2 != Float 2.0 inside a value
    v = np.arange(4, dtype=np.float32)
    assert _compare(v, "==", v.copy()) and _compare(v, "!=", 0)


def test_numbers():
    assert _compare(1, "==", 1.0) and not _compare(1, "!=", 1.0)                      # mixed pairs as f64 at the top level
    assert _compare(float("nan"), "==", float("nan"))                                  # Float/Float: total order
    assert _compare(-0.0, "!=", 0.0) and _compare(-0.0, "<", 0.0)
    assert _compare(2, "<", 3) and _compare(2.5, ">=", 2) and _compare(np.int64(7), "<=", np.float64(7.0))


@pytest.mark.parametrize("op", ["<", "<=", ">", ">="])
def test_ordering_operators_keep_the_type_error(op):
    for a, b in ((None, 5), ("a", 5), (True, 1), (np.zeros(2, np.float32), 0)):
        with pytest.raises(TypeError):
            _compare(a, op, b)
