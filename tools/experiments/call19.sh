cd $GRAFT_REPO_ROOT
tools/exp_ab.sh c19 tree:1 fs9216:1 fs12288:1 fs6144:1 fs192:1 fs3:1 fs1:1 tree:4 fs9216:4 fs6144:4 tree:1:hd1080 fl8192:1:hd1080 fl6144:1:hd1080 fl10240:1:hd1080 fl512:1:hd1080 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$10,$11}'
