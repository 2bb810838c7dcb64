#!/bin/bash
# Pins the oracle to the reference: builds oracle/ref_fixtures against the real cozo-core and writes
# tests/golden/ref_fixtures.json, which tests/test_ref_fixtures.py compares the oracle (and through it the device path)
# with.  Needs cargo + rustc and the crates cozo-core depends on (crates.io or a vendored registry); neither exists in
# the image this repository was developed in, so the fixture file is absent there and the test skips with that reason.
#   COZO_CORE_PATH=/path/to/cozo/cozo-core oracle/ref_fixtures/make_ref_fixtures.sh
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
CORE=${COZO_CORE_PATH:-/root/reference/cozo-core}
command -v cargo >/dev/null || { echo "cargo not found: the reference cannot be built on this box" >&2; exit 3; }
[ -f "$CORE/Cargo.toml" ] || { echo "no cozo-core at $CORE (set COZO_CORE_PATH)" >&2; exit 3; }
WORK=$ROOT/oracle/_ref/ref_fixtures
mkdir -p "$WORK/src"
sed "s|@COZO_CORE_PATH@|$CORE|" "$HERE/Cargo.toml" > "$WORK/Cargo.toml"
cp "$HERE/src/main.rs" "$WORK/src/main.rs"
python3 "$ROOT/tests/golden/make_ref_inputs.py" "$WORK/inputs.json"
(cd "$WORK" && CARGO_TARGET_DIR="$ROOT/oracle/_ref/target" cargo build --release)
"$ROOT/oracle/_ref/target/release/cozo_ref_fixtures" "$WORK/inputs.json" "$ROOT/tests/golden/ref_fixtures.json"
# PageRank again on ONE rayon thread: graph 0.3.1's loop is deterministic there under either reading of its contribution refresh
# (tests/test_ref_fixtures.py::check_pagerank tells the readings apart from these rows and the default-thread rows above)
python3 "$ROOT/tests/golden/make_ref_inputs.py" --pagerank-only "$WORK/inputs_pagerank.json"
RAYON_NUM_THREADS=1 "$ROOT/oracle/_ref/target/release/cozo_ref_fixtures" "$WORK/inputs_pagerank.json" "$ROOT/tests/golden/ref_fixtures_pagerank_1thread.json"
echo "wrote $ROOT/tests/golden/ref_fixtures.json and ref_fixtures_pagerank_1thread.json; now run: python -m pytest tests/test_ref_fixtures.py -q -s"
