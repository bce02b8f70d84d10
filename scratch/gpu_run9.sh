#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CFM_PROFILE=1 CFM_LIB_OVERRIDE=$PWD/scratch/variants/prof.so timeout 300 python scratch/asg_one.py C3 2 2>&1 | tail -5
