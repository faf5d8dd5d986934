cd $GRAFT_REPO_ROOT
python tools/gp_phases.py 2>&1 | grep -v "^{" | tail -20
