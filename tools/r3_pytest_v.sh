#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -x -v 2>&1 | grep -v "^  File\|Extension modules" > gpurun_out/pytest_verbose.txt; tail -25 gpurun_out/pytest_verbose.txt | cut -c1-220
