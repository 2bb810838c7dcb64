#!/bin/bash
# the round-end checks as the driver runs them: the whole gpu suite, then smoke
O=gpurun_out/r2j; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.txt
