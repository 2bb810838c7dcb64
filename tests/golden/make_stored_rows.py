"""Regenerates tests/golden/stored_rows.json: the byte forms of a few rows as cozo_amd/codec.py writes them today (hex).
These pin OUR encoders against accidental change; they are not reference output (no cargo here -- see codec.py's header for what
the formats restate).  Run from the repo root: python tests/golden/make_stored_rows.py"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cozo_amd import codec  # noqa: E402

ROWS = [
    ("ints", [2095, -1, 0, 2 ** 53, -(2 ** 63)], 5),
    ("floats", [1.0, -0.0, 2.5, float("inf"), float("-inf")], 5),
    ("strings", ["MSS", "", "abcdefgh", "abcdefghi", "ü"], 5),
    ("mixed-key-and-value", [7, "k", None, True, False, b"\x00\x01", [1, "x", [2.5]], 3.25, "value", -70000, 2 ** 40], 2),
    ("vector-row", ["doc-1", 3, np.array([1.5, -2.0, 0.25], dtype=np.float32), [np.array([1.0], dtype=np.float32), "x"]], 1),
]


def build():
    out = {}
    for name, row, n_key in ROWS:
        out[name] = {"n_key_cols": n_key,
                     "key": codec.encode_key_for_store(9, row[:n_key]).hex(),
                     "val": codec.encode_val_for_store(9, row[n_key:]).hex()}
    # a two-node index through the native write-back encoder (czi_hnsw_encode_rows): digest of all key / value bytes
    from cozo_amd import build as B
    from cozo_amd.ingest import encode_index_rows
    B.build_ingest()
    vecs = np.array([[1, 2], [2, 3]], dtype=np.float32)
    nb = [np.array([[1, 0xFFFFFFFF], [0, 0xFFFFFFFF]], dtype=np.uint32)]
    rows = encode_index_rows([("a", 1, -1), ("b", 1, -1)], vecs, [None], nb, 0, 0, [np.array([[2.0, 0], [2.0, 0]])], 5)
    out["index-two-nodes"] = {"rows": len(rows), "keys_sha256": hashlib.sha256(rows.keys).hexdigest(),
                              "vals_sha256": hashlib.sha256(rows.vals).hexdigest(), "first_key": rows.row(0)[0].hex(),
                              "first_val": rows.row(0)[1].hex()}
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stored_rows.json")
    json.dump(build(), open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
