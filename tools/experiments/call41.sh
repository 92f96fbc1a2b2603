#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/exp_ab.sh ab41 c5:1 rz6:1 rz5:1 tree:1 c5:1:hd1080 rz6:1:hd1080 rz5:1:hd1080
