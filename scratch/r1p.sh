#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
