#!/bin/bash
# the full GPU suite twice + smoke (flakiness check of the final build)
cd $GRAFT_REPO_ROOT
for i in 1 2; do python -m pytest tests -q -m gpu 2>&1 | tail -3 | cut -c1-200; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
