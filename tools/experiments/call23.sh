cd $GRAFT_REPO_ROOT
tools/exp_ab.sh c23 tree:1 bs5:1 bs6:1 bs12:1 tree:1 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$18,$19}'
