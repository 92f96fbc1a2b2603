cd $GRAFT_REPO_ROOT
tools/exp_ab.sh c20 tree:1 fs1:1 fs1b:1 fs1c:1 fs1d:1 fs1e:1 fs1f:1 tree:1 fs1:1 fs1:4 tree:4 fs1:0 tree:0 tree:1:hd1080 fl1:1:hd1080 fl1b:1:hd1080 fl1c:1:hd1080 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$10,$11}'
