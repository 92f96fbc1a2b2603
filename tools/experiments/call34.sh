#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/exp_ab.sh ab34 tree:1 abl1:1 abl2:1 abl4:1 abl8:1 abl16:1 tree:1
