#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
bash scripts/gpu/issue_profile.sh r5issue 2>&1 | tail -150
bash scripts/gpu/cprof.sh
